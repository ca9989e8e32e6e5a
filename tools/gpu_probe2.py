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
      ms = eng.time_step(acts[:100], abi.STEP_PHYSICS)
      z = eng.state[:, 2].mean().item()
      print(f'{dtype} N={n} iters={iters}: {ms*1e3:.1f} us/launch  mean base z {z:.3f}', flush=True)
      eng.close()
# average number of touching spheres per robot in the benchmark regime (CPU oracle FK on a sample)
from oracle import solo_oracle as so
ca, ma = make_abi('float32')
eng = Engine(ca, ma, 4096)
g = torch.Generator(device='cuda').manual_seed(1234)
acts = (torch.rand(300, 4096, 12, device='cuda', dtype=torch.float32, generator=g) * 2 - 1) * (2 * np.pi)
eng.rollout(acts, abi.STEP_PHYSICS)
st = eng.state.cpu().numpy().astype(np.float64)
ph = so.OraclePhysics(ca, ma)
rad = np.array(list(ma.sphere_radius))
cnt = []
for e in range(512):
  c = ph.sphere_centers(st[e])
  cnt.append(int(((c[:, 2] - rad) < ca.contact_margin).sum()))
print('touching spheres per robot: mean %.2f  max %d  hist %s' % (np.mean(cnt), max(cnt), np.bincount(cnt).tolist()), flush=True)
