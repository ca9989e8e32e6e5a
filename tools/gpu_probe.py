"""Quick GPU probe: timing of the step kernel at several batch sizes (not a test)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch
from gym_solo_amd import abi
from gym_solo_amd.engine import Engine
from helpers import make_abi

print('torch', torch.__version__, torch.cuda.get_device_name(0), flush=True)
for dtype, tdt in (('float32', torch.float32), ('float64', torch.float64)):
  for n in (1024, 4096, 16384):
    ca, ma = make_abi(dtype)
    t0 = time.time()
    eng = Engine(ca, ma, n)
    t_create = time.time() - t0
    g = torch.Generator(device='cuda').manual_seed(1234)
    acts = (torch.rand(64, n, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * (2 * np.pi)
    for i in range(64):
      eng.step(acts[i], abi.STEP_PHYSICS)
    ms = eng.time_step(acts[:100], abi.STEP_PHYSICS)
    print(f'{dtype} N={n}: create+settle {t_create:.2f}s  {ms*1e3:.1f} us/launch  '
          f'{n/ms*1e3:.3e} env-steps/s', flush=True)
    eng.close()
