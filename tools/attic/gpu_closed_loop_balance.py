"""MEASUREMENT: one launch per env step (the closed loop) with the launch order re-balanced every M steps (Engine.balance():
costliest robots first, by the sweeps of their last step; the argsort and the order upload are INSIDE the timed region).
  python tools/gpu_closed_loop_balance.py float64 0 8 32 128"""
import sys, os, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from gym_solo_amd import abi

dtype = sys.argv[1] if len(sys.argv) > 1 else 'float64'
periods = [int(x) for x in sys.argv[2:]] or [0, 8, 32, 128]
n, steps = 4096, 512
tdt = torch.float32 if dtype == 'float32' else torch.float64
env = bench.build_env(n, 0, dtype, steps_per_launch=1, rollout_streams=1, migrate_steps=0)
eng = env.engine
gen = torch.Generator(device='cuda').manual_seed(1234)
bench.desynchronise_episodes(eng, gen)
acts = (torch.rand(steps, n, abi.NUM_JOINTS, device='cuda', dtype=tdt, generator=gen) * 2 - 1) * 6.283185307179586
for m in periods + periods[:1]:
  eng.set_order(None)
  ts = []
  for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
      if m and i % m == 0: eng.balance()
      eng.step(acts[i], abi.STEP_ALL)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
  print('%s closed loop, order re-balanced every %3d steps: %.4g env-steps/s (median of 5 x %d steps)' % (dtype, m, n * steps / statistics.median(ts), steps), flush=True)
env._close()
