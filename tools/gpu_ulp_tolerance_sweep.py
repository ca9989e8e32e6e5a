"""MEASUREMENT: what the Gauss-Seidel convergence tolerance (SoloConfig.solver_ulp_tolerance: a row whose clamped candidate differs
from its impulse by at most k half-ulps, relative, is left alone; default 2) costs and buys - throughput of the driver's 20-step
launch and of one launch per step, mean sweeps per robot-step, and the error against the oracle (which always runs its 50
sweeps) over 60 steps of the benchmark's random actions.   gpurun -- python tools/gpu_ulp_tolerance_sweep.py"""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import bench
from gym_solo_amd import abi
from gym_solo_amd.engine import Engine
from helpers import make_abi, random_actions
from oracle import solo_oracle as so

TOLS = [int(x) for x in (sys.argv[1:] or ['0', '2', '8', '32', '128', '512', '4096'])]
n = 4096
for tol in TOLS:
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig, Solo8VanillaEnv
  from gym_solo_amd.workloads import register_benchmark_workload
  out = {}
  for leg in ('k20', 'closed'):
    closed = leg == 'closed'
    cfg = Solo8VanillaConfig()
    cfg.num_envs, cfg.dtype, cfg.auto_reset, cfg.solver_ulp_tolerance = n, 'float64', True, tol
    if closed:
      cfg.steps_per_launch, cfg.rollout_streams, cfg.migrate_steps = 1, 1, 0
    env = Solo8VanillaEnv(config=cfg, copy_outputs=False)
    register_benchmark_workload(env, max_steps=1000)
    env._ensure_program()
    eng = env.engine
    gen = torch.Generator(device='cuda').manual_seed(1234)
    bench.desynchronise_episodes(eng, gen)
    pool = lambda k: (torch.rand(k, n, 12, device='cuda', dtype=torch.float64, generator=gen) * 2 - 1) * 6.283185307179586
    bufs = None if closed else eng.rollout_buffers(20)
    def run(a):
      if closed:
        for i in range(a.shape[0]):
          eng.step(a[i], abi.STEP_ALL)
      else:
        eng.rollout(a, abi.STEP_ALL, out=bufs)
    run(pool(20))
    ts, sweeps = [], []
    for _ in range(40):
      a = pool(20)
      torch.cuda.synchronize(); t0 = time.perf_counter(); run(a); torch.cuda.synchronize()
      ts.append(time.perf_counter() - t0)
      if not closed:
        sweeps.append(float(eng.cost.double().mean()) / 20)
    out[leg] = n * 20 / statistics.median(ts)
    if not closed:
      out['sweeps'] = statistics.mean(sweeps)
    env._close()
  # parity: 60 steps of U(-2 pi, 2 pi) targets, 512 robots, against the oracle's 50 plain sweeps
  ca, ma = make_abi('float64', steps_per_launch=60, solver_ulp_tolerance=tol)
  e = Engine(ca, ma, 512)
  ph = so.OraclePhysics(ca, ma)
  st = e.state.cpu().numpy().copy()
  rng = np.random.default_rng(99)
  a = np.stack([random_actions(rng, 512) for _ in range(60)])
  e.rollout(torch.as_tensor(a, device='cuda'), abi.STEP_PHYSICS)
  for k in range(60):
    ph.step(st, a[k], threads=16)
  err = np.abs(e.state.cpu().numpy()[:, :29] - st[:, :29]).max(axis=1)
  # ... and a robot at rest: 1000 zero-target steps from the settled pose, the joint rates it keeps
  zero = torch.zeros(1000, 512, 12, device='cuda', dtype=torch.float64)
  e.reset(None)
  e.rollout(zero, abi.STEP_PHYSICS)
  rest = float(e.state[:, abi.S_QD:abi.S_QD + 8].abs().max())
  e.close()
  print('solver_ulp_tolerance %5d: K = 20 %.4g env-steps/s, one launch per step %.4g, mean sweeps per robot-step %.2f; 60 flailing steps vs oracle: max %.1e, p99 %.1e; '
        'joint rates after 1000 zero-target steps %.1e rad/s' % (tol, out['k20'], out['closed'], out['sweeps'], err.max(), np.quantile(err, 0.99), rest), flush=True)
