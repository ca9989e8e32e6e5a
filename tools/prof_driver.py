"""Runs a short physics+obs+reward rollout for profiling under rocprofv3 (not a test)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
sys.argv += [''] * 4
iters = int(sys.argv[1] or 50); dtype = sys.argv[2] or 'float32'; n = int(sys.argv[3] or 4096)
from bench import build_env
from gym_solo_amd import abi
env = build_env(n, 0, dtype, steps_per_launch=int(os.environ.get('SPL', '100')), rollout_streams=int(os.environ.get('STREAMS', '2')))
if iters != 50:
  raise SystemExit('use the config default')
eng = env.engine
tdt = torch.float32 if dtype == 'float32' else torch.float64
g = torch.Generator(device='cuda').manual_seed(1234)
acts = (torch.rand(600, n, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * (2 * np.pi)
out = eng.rollout_buffers(acts.shape[0])
eng.rollout(acts, abi.STEP_ALL, out=out)
torch.cuda.synchronize()
print('done', eng.kernel_name)
