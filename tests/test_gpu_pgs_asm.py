"""The assembly Gauss-Seidel loops (gym_solo_amd/csrc/solo_pgs_gfx950.h: f32 and, since round 3, f64) against
their C++ definition, ON THE GPU and BIT FOR BIT: the product library and libsolo_hip_pgs_cpp.so (the same
translation unit built with -DSOLO_PGS_NO_ASM) run the same contact-rich rollouts in two processes; states,
rewards, done flags and per-robot sweep counts must be identical - the assembly takes the same rows in the
same order with the same arithmetic, and stops after the same number of sweeps."""
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
from gym_solo_amd.engine import Engine
from helpers import make_abi
case, out, dtype = sys.argv[1], sys.argv[2], sys.argv[3]
tdt = torch.float32 if dtype == 'float32' else torch.float64
res = {}
if case == 'flail':      # the bench workload: random targets, robots tumbling over the plane, auto-reset
  from bench import build_env
  env = build_env(4096, 0, dtype, steps_per_launch=250, rollout_streams=2)   # the bench geometry
  eng = env.engine
  g = torch.Generator(device='cuda').manual_seed(77)
  acts = (torch.rand(500, 4096, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * (2 * np.pi)
  o = eng.rollout_buffers(500)
  eng.rollout(acts, abi.STEP_ALL, out=o)
  torch.cuda.synchronize()
  res = dict(state=eng.state.cpu().numpy(), cost=eng.cost.cpu().numpy(),
             reward=o[1].cpu().numpy(), done=o[2].cpu().numpy(), obs=o[0].cpu().numpy())
else:                     # few sweeps allowed / exact tolerance / one sweep: the loop's exits
  # ('resid': pybullet's solverResidualThreshold 1e-7 - the copy of the loop with the residual test at the end of a sweep)
  iters, tol, resid = {'cap3': (3, 2, 0.0), 'exact': (50, 0, 0.0), 'one': (1, 2, 0.0), 'resid': (50, 2, 1e-7), 'resid_cap': (4, 2, 1e-9)}[case]
  ca, ma = make_abi(dtype, solver_iterations=iters, solver_ulp_tolerance=tol, settle_steps=100, solver_residual_threshold=resid)
  eng = Engine(ca, ma, 256)
  rng = np.random.default_rng(5)
  acts = torch.as_tensor(rng.uniform(-6, 6, (120, 256, 12)), device='cuda', dtype=tdt)
  for i in range(120):
    eng.step(acts[i], abi.STEP_PHYSICS)
  torch.cuda.synchronize()
  res = dict(state=eng.state.cpu().numpy(), cost=eng.cost.cpu().numpy())
np.savez(out, **res)
'''


def _run(lib, case, dtype, tmp_path):
  out = str(tmp_path / ('%s_%s_%s.npz' % (case, dtype, os.path.basename(lib))))
  env = dict(os.environ, SOLO_HIP_LIB=lib)
  subprocess.run([sys.executable, '-c', _WORKER % {'root': ROOT}, case, out, dtype], check=True, env=env, timeout=600)
  return np.load(out)


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
@pytest.mark.parametrize('case', ['flail', 'cap3', 'exact', 'one', 'resid', 'resid_cap'])
def test_assembly_loop_equals_cpp_loop_bit_for_bit(case, dtype, tmp_path):
  asm_lib, cpp_lib = os.path.join(CSRC, 'libsolo_hip.so'), os.path.join(CSRC, 'libsolo_hip_pgs_cpp.so')
  assert os.path.isfile(cpp_lib), 'build it: make -C gym_solo_amd/csrc test-libs (or __graft_entry__.build())'
  a, b = _run(asm_lib, case, dtype, tmp_path), _run(cpp_lib, case, dtype, tmp_path)
  assert set(a.files) == set(b.files)
  for k in a.files:
    assert a[k].shape == b[k].shape
    assert a[k].tobytes() == b[k].tobytes(), 'assembly and C++ Gauss-Seidel loops differ in %r' % k
  assert a['cost'].max() > 0  # (sweeps were counted at all)
