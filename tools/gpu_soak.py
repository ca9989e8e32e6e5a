"""Soak run (not a test): many episodes of the bench workload; prints the episodic statistics and
checks the invariants (finite states, unit quaternions, no restored robots, episode accounting)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from gym_solo_amd import abi
from gym_solo_amd.distributed import summarize
from bench import build_env
for dtype, episodes in (('float32', 20), ('float64', 3)):
  n = 4096
  env = build_env(n, 0, dtype, steps_per_launch=250, rollout_streams=2)
  eng = env.engine
  tdt = torch.float32 if dtype == 'float32' else torch.float64
  g = torch.Generator(device='cuda').manual_seed(2024)
  for ep in range(episodes):
    acts = (torch.rand(1001, n, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * (2 * np.pi)
    eng.rollout(acts, abi.STEP_ALL)
  torch.cuda.synchronize()
  st = eng.state
  s = summarize(eng.stats.cpu().numpy())
  qn = st[:, abi.S_QUAT:abi.S_QUAT + 4].norm(dim=1)
  print(dtype, s, 'finite', bool(torch.isfinite(st).all()), 'quat norm err %.2e' % float((qn - 1).abs().max()),
        'z min/max %.3f %.3f' % (float(st[:, 2].min()), float(st[:, 2].max())))
  assert s['episodes'] == n * episodes and s['diverged'] == 0 and s['mean_length'] == 1001.0
  env._close()
print('soak ok')
