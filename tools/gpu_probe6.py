import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from gym_solo_amd import abi
from bench import build_env
n = 4096
env = build_env(n, 0, 'float32'); eng = env.engine
g = torch.Generator(device='cuda').manual_seed(1234)
acts = (torch.rand(1000, n, 12, device='cuda', dtype=torch.float32, generator=g) * 2 - 1) * (2 * np.pi)
eng.rollout(acts[:100], abi.STEP_ALL); torch.cuda.synchronize()
for rep in range(2):
  t0 = time.perf_counter(); eng.rollout(acts, abi.STEP_ALL); torch.cuda.synchronize(); t2 = time.perf_counter()
  print(f'rollout 1000 (steps_per_launch={eng.steps_per_launch}): {(t2-t0)*1e3:.1f} us/step', flush=True)
