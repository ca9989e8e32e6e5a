import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from gym_solo_amd import abi
from bench import build_env
n=4096
env = build_env(n, 0, 'float32', steps_per_launch=250, rollout_streams=2)
eng = env.engine
g = torch.Generator(device='cuda').manual_seed(1234)
acts = (torch.rand(1000, n, 12, device='cuda', dtype=torch.float32, generator=g) * 2 - 1) * (2 * np.pi)
for name, flags in (('ALL', abi.STEP_ALL), ('PHYSICS|DONE', abi.STEP_PHYSICS | abi.STEP_DONE), ('PHYSICS', abi.STEP_PHYSICS)):
  eng.rollout(acts[:250], flags); torch.cuda.synchronize()
  best = 1e9
  for rep in range(2):
    t0 = time.perf_counter(); eng.rollout(acts, flags); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
  print(f'{name}: {best*1e3:.2f} us/step  {n*1000/best:.3e} env-steps/s', flush=True)
out = eng.rollout_buffers(1000)
for rep in range(3):
  t0 = time.perf_counter(); eng.rollout(acts, abi.STEP_ALL, out=out); torch.cuda.synchronize(); dt = time.perf_counter() - t0
  print(f'ALL recorded, pass {rep}: {dt*1e3:.2f} us/step  {n*1000/dt:.3e} env-steps/s', flush=True)
