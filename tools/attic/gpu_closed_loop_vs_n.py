import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch
from bench import build_env
from gym_solo_amd import abi
for n in (256, 1024, 2048, 4096, 8192):
  env = build_env(n, 0, 'float32', steps_per_launch=1, rollout_streams=1)
  eng = env.engine
  g = torch.Generator(device='cuda').manual_seed(1234)
  acts = (torch.rand(600, n, 12, device='cuda', generator=g) * 2 - 1) * (2 * np.pi)
  for i in range(300): eng.step(acts[i], abi.STEP_ALL)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for i in range(300, 600): eng.step(acts[i], abi.STEP_ALL)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 300
  cost = eng.cost.cpu().numpy()
  print('N=%5d: %.1f us/step -> %.3e env-steps/s ; sweeps last step: mean %.1f max %d, at cap %d' % (n, dt * 1e6, n / dt, cost.mean(), cost.max(), (cost >= 50).sum()), flush=True)
  env._close()
