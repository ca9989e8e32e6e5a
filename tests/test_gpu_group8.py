"""EXPERIMENT build `make -C gym_solo_amd/csrc group8` (one wave computes the leg dynamics of the 8 robots of a
workgroup: gym_solo_amd/csrc/solo_step_kernel_g8.h) against the one-wave-per-robot kernel, ON THE GPU and BIT FOR
BIT: two libraries run the same contact-rich rollouts in two processes; states, observations, rewards, done flags,
per-robot sweep counts and the settle snapshot must be identical - the dynamics wave evaluates the product's
expressions under another lane mapping, with the per-leg sums in the same association.  Both sides of the comparison
are built with floating-point contraction OFF (libsolo_hip_nocontract.so / libsolo_hip_group8_nocontract.so): under
-ffp-contract=fast the compiler picks different multiply-add fusions in the two code shapes, a rounding-level
difference in EITHER build's favour, not a difference in the arithmetic being compared.  The measurement the build
exists for is tools/ab_group8.sh (profiles/round3_group8_ab.log; product flags on both sides).  Skipped when the
libraries were not built (they are not part of `make all`; __graft_entry__.build() makes them)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'gym_solo_amd', 'csrc')

_WORKER = r'''
import sys, os
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import numpy as np, torch
from gym_solo_amd import abi
from bench import build_env
out, dtype, n, spl, streams, terrain = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
tdt = torch.float32 if dtype == 'float32' else torch.float64
env = build_env(n, 0, dtype, max_steps=150, steps_per_launch=spl, rollout_streams=streams)
eng = env.engine
if terrain != 'flat':
  import helpers
  eng.set_terrain(getattr(helpers, terrain + '_terrain')())
g = torch.Generator(device='cuda').manual_seed(77)
rng = np.random.default_rng(5)
eng.set_params(0, torch.as_tensor(rng.uniform(0.3, 1.0, n), device='cuda', dtype=tdt).contiguous())
eng.set_params(1, torch.as_tensor(rng.uniform(0.8, 1.2, n), device='cuda', dtype=tdt).contiguous())
eng.settle()
k = 400
acts = (torch.rand(k, n, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * (2 * np.pi)
o = eng.rollout_buffers(k)
eng.rollout(acts, abi.STEP_ALL, out=o)
torch.cuda.synchronize()
np.savez(out, state=eng.state.cpu().numpy(), cost=eng.cost.cpu().numpy(), reward=o[1].cpu().numpy(), done=o[2].cpu().numpy(),
         obs=o[0].cpu().numpy(), stats=eng.stats.cpu().numpy()[[2, 3, 5]], snapshot=eng.snapshot.cpu().numpy())
'''


def _run(lib, tag, args, tmp_path):
  out = str(tmp_path / ('%s_%s.npz' % (tag, os.path.basename(lib))))
  env = dict(os.environ, SOLO_HIP_LIB=lib)
  subprocess.run([sys.executable, '-c', _WORKER % {'root': ROOT}, out] + [str(a) for a in args], check=True, env=env, timeout=600)
  return np.load(out)


@pytest.mark.parametrize('dtype,n,spl,streams,terrain', [('float32', 1024, 100, 2, 'flat'), ('float32', 256, 1, 1, 'flat'),
                                                         ('float32', 512, 20, 1, 'stairs'), ('float64', 512, 50, 1, 'flat')])
def test_group_of_eight_build_equals_the_product_bit_for_bit(dtype, n, spl, streams, terrain, tmp_path):
  prod, g8 = os.path.join(CSRC, 'libsolo_hip_nocontract.so'), os.path.join(CSRC, 'libsolo_hip_group8_nocontract.so')
  if not (os.path.isfile(g8) and os.path.isfile(prod)):
    pytest.skip('experiment library not built: make -C gym_solo_amd/csrc group8')
  tag = '%s_%d_%d_%d_%s' % (dtype, n, spl, streams, terrain)
  a, b = _run(prod, tag, (dtype, n, spl, streams, terrain), tmp_path), _run(g8, tag, (dtype, n, spl, streams, terrain), tmp_path)
  assert set(a.files) == set(b.files)
  for k in a.files:
    assert a[k].shape == b[k].shape
    assert a[k].tobytes() == b[k].tobytes(), 'group-of-8 build and product differ in %r (max |diff| %.3g)' % (
      k, float(np.nanmax(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)))))
  assert a['cost'].max() > 0 and a['stats'][0] > 0   # sweeps were counted, episodes ended
