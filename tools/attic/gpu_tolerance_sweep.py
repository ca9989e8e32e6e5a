"""Sweeps to convergence and rollout throughput vs the solver's early-exit tolerance (solver_ulp_tolerance:
impulse changes of <= k half-ulps, relative, count as converged; 0 = exact), f32 bench workload.
Diagnostic: the default (2) is the f32 rounding level; this shows what looser settings would buy."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gym_solo_amd import abi
from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig, Solo8VanillaEnv
from gym_solo_amd.workloads import register_benchmark_workload
n = 4096
for tol in (0, 2, 4, 8, 16, 64):
  cfg = Solo8VanillaConfig()
  cfg.num_envs, cfg.device, cfg.dtype, cfg.auto_reset = n, 0, 'float32', True
  cfg.steps_per_launch, cfg.rollout_streams, cfg.solver_ulp_tolerance = 250, 2, tol
  env = Solo8VanillaEnv(config=cfg, copy_outputs=False)
  register_benchmark_workload(env, max_steps=1000)
  env._ensure_program()
  eng = env.engine
  g = torch.Generator(device='cuda').manual_seed(1234)
  acts = (torch.rand(1500, n, 12, device='cuda', generator=g) * 2 - 1) * (2 * np.pi)
  out = eng.rollout_buffers(1500)
  eng.rollout(acts, abi.STEP_ALL, out=out)
  torch.cuda.synchronize()
  ts = []
  for rep in range(3):
    t0 = time.perf_counter(); eng.rollout(acts, abi.STEP_ALL, out=out); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
  cost = eng.cost.cpu().numpy() / 250.0   # sweeps per step in the last launch
  print('ulp_tol %3d: %.3e env-steps/s ; sweeps per step mean %.2f p50 %.2f p99 %.2f max %.2f' % (
      tol, 1500 * n / np.median(ts), cost.mean(), np.median(cost), np.percentile(cost, 99), cost.max()), flush=True)
  env._close()
