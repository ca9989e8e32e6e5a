"""MEASUREMENT: throughput of BASELINE configs[3] (8192 envs/GPU, per-env friction + base-mass
randomisation) and configs[4] (4096 envs/GPU on the incline / stairs heightfields), same rollout path
as bench.py (f32, 250 steps per launch, 2 slices, every step recorded)."""
import sys, os, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import helpers
from gym_solo_amd import abi
from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig, Solo8VanillaEnv
from gym_solo_amd.workloads import register_benchmark_workload

def run(name, n, terrain=None, randomise=False, k=1500):
  cfg = Solo8VanillaConfig()
  cfg.num_envs, cfg.dtype, cfg.auto_reset, cfg.steps_per_launch, cfg.rollout_streams = n, 'float32', True, 250, 2
  if terrain is not None:
    cfg.terrain = terrain
  env = Solo8VanillaEnv(config=cfg, copy_outputs=False)
  register_benchmark_workload(env, max_steps=1000)
  env._ensure_program()
  eng = env.engine
  g = torch.Generator(device='cuda').manual_seed(4321)
  if randomise:
    eng.set_params(abi.PARAM_FRICTION, torch.rand(n, device='cuda', generator=g) * 0.7 + 0.3)
    eng.set_params(abi.PARAM_BASE_MASS_SCALE, torch.rand(n, device='cuda', generator=g) * 0.4 + 0.8)
    eng.settle()
  acts = (torch.rand(k, n, 12, device='cuda', generator=g) * 2 - 1) * (2 * np.pi)
  out = eng.rollout_buffers(k)
  eng.rollout(acts[:250], abi.STEP_ALL)
  ts = []
  for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.rollout(acts, abi.STEP_ALL, out=out)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
  st = eng.stats.cpu().numpy()
  print('%-58s %.3g env-steps/s (median of 5 x %d steps; diverged %d)' % (name, n * k / statistics.median(ts), k, st[5]), flush=True)
  env._close()

run('configs[1]  4096 envs, flat plane', 4096)
run('configs[3]  8192 envs, friction U(.3,1) + base mass U(.8,1.2)', 8192, randomise=True)
run('configs[4]  4096 envs, 10 degree incline heightfield', 4096, terrain=helpers.incline_terrain())
run('configs[4]  4096 envs, stairs 0.03 m x 0.30 m heightfield', 4096, terrain=helpers.stairs_terrain())
