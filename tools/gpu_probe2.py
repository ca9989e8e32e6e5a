"""GPU probe: fixed vs per-iteration cost of the step kernel (not a test)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from gym_solo_amd import abi
from gym_solo_amd.engine import Engine
from helpers import make_abi
for dtype, tdt in (('float32', torch.float32),):
  for n in (4096,):
    for iters in (1, 10, 50, 100):
      ca, ma = make_abi(dtype, solver_iterations=iters)
      eng = Engine(ca, ma, n)
      g = torch.Generator(device='cuda').manual_seed(1234)
      acts = (torch.rand(128, n, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * (2 * np.pi)
      eng.rollout(acts, abi.STEP_PHYSICS)
      # average touching spheres per env at this point
      ms = eng.time_step(acts[0], abi.STEP_PHYSICS, reps=200)
      z = eng.state[:, 2].mean().item()
      print(f'{dtype} N={n} iters={iters}: {ms*1e3:.1f} us/launch  mean base z {z:.3f}', flush=True)
      eng.close()
