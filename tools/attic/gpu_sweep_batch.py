"""GPU probe: kernel time vs batch size and flags (not a test)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from gym_solo_amd import abi
from bench import build_env
for dtype, tdt in (('float32', torch.float32), ('float64', torch.float64)):
  for n in (256, 1024, 2048, 4096, 8192, 16384):
    env = build_env(n, 0, dtype)
    eng = env.engine
    g = torch.Generator(device='cuda').manual_seed(1234)
    acts = (torch.rand(128, n, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * (2 * np.pi)
    eng.rollout(acts, abi.STEP_ALL)
    ms_all = eng.time_step(acts[:100], abi.STEP_ALL)
    ms_phy = eng.time_step(acts[:100], abi.STEP_PHYSICS)
    ms_red = eng.time_step(None, abi.STEP_OBS | abi.STEP_REWARD, reps=200)
    print(f'{dtype} N={n}: all {ms_all*1e3:.1f} us  physics {ms_phy*1e3:.1f} us  obs+reward only {ms_red*1e3:.1f} us  -> {n/ms_all*1e3:.3e} env-steps/s', flush=True)
    env._close()
