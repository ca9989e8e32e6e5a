"""MEASUREMENT: is a longest-first order of the robots, from the cost the engine keeps (view.cost = sweeps
of the robot's last launch), worth anything?  Same process, alternating: every rollout preceded (outside the timed region) by
Engine.balance() - costliest robots first, in the dispatch order and in the migration rings - or by set_order(None).
N = 4096 (every robot on a wave slot of its own: the order only permutes the dispatch) and N = 8192 (two rounds, robot
migration: the order decides who starts first), K = 20 and K = 1000; f64.  What the order could use is what
tools/gpu_cost_persistence.py measures: rank correlation 0.14 between consecutive 20-step launches.
  python tools/gpu_order_ab.py"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gym_solo_amd import abi

for n in (4096, 8192):
  for k, reps in ((20, 40), (1000, 6)):
    env = bench.build_env(n, 0, 'float64')
    eng = env.engine
    gen = torch.Generator(device='cuda').manual_seed(1234)
    bench.desynchronise_episodes(eng, gen)
    out = eng.rollout_buffers(k)
    pool = lambda: (torch.rand(k, n, abi.NUM_JOINTS, device='cuda', dtype=torch.float64, generator=gen) * 2 - 1) * 6.283185307179586
    eng.rollout(pool(), abi.STEP_ALL, out=out)
    times = {'identity': [], 'longest first': []}
    for rep in range(2 * reps):
      mode = 'longest first' if rep % 2 else 'identity'
      a = pool()
      if mode == 'longest first':
        eng.balance()
      else:
        eng.set_order(None)
      torch.cuda.synchronize(); t0 = time.perf_counter()
      eng.rollout(a, abi.STEP_ALL, out=out)
      torch.cuda.synchronize(); times[mode].append(time.perf_counter() - t0)
    p = eng.plan(k)
    print('N = %d, K = %d (%d x %d steps, %d slice(s), migrate %d): identity order %.4g env-steps/s, longest first (by the last launch\'s sweeps) %.4g (%+.1f %%)' % (
      n, k, p['launches'], p['steps_per_launch'], p['slices'], p['migrate_steps'], n * k / statistics.median(times['identity']),
      n * k / statistics.median(times['longest first']), 100.0 * (statistics.median(times['identity']) / statistics.median(times['longest first']) - 1.0)), flush=True)
    env._close()
