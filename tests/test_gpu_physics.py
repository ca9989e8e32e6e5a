"""GPU parity of the physics path (A3 + A4) through the C-ABI vs the CPU oracle."""
import numpy as np
import pytest

from gym_solo_amd import abi
from helpers import make_abi, random_actions

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def torch():
  import torch
  if not torch.cuda.is_available():
    pytest.fail('GPU tests need a visible MI355X')
  return torch


def test_zero_copy_views(torch):
  from gym_solo_amd.engine import Engine
  ca, ma = make_abi('float32', settle_steps=3)
  eng = Engine(ca, ma, 8)
  assert eng.state.shape == (8, abi.STATE_STRIDE) and eng.state.dtype == torch.float32
  assert eng.state.is_cuda
  before = eng.state.clone()
  eng.step(torch.zeros(8, 12, device='cuda'), abi.STEP_PHYSICS)
  eng.synchronize()
  assert not torch.equal(before, eng.state)  # the view aliases engine memory
  eng.close()


def test_settle_matches_oracle_f64(torch):
  """Contractive scenario (robot drops and folds, solo8v2vanilla.py:124-136): 500 steps."""
  from gym_solo_amd.engine import Engine
  from oracle import solo_oracle as so
  ca, ma = make_abi('float64')
  eng = Engine(ca, ma, 4)
  ref = so.OraclePhysics(ca, ma).settle(1)
  got = eng.snapshot.cpu().numpy()
  np.testing.assert_allclose(got[:, :29], np.tile(ref[:, :29], (4, 1)), rtol=0, atol=1e-9)
  eng.close()


def test_random_rollout_matches_oracle_f64(torch):
  from gym_solo_amd.engine import Engine
  from oracle import solo_oracle as so
  ca, ma = make_abi('float64')
  n, steps = 64, 60
  eng = Engine(ca, ma, n)
  ph = so.OraclePhysics(ca, ma)
  st = eng.state.cpu().numpy().copy()
  rng = np.random.default_rng(0)
  for k in range(steps):
    a = random_actions(rng, n)
    ph.step(st, a)
    eng.step(torch.as_tensor(a, device='cuda'), abi.STEP_PHYSICS)
  got = eng.state.cpu().numpy()
  # contact-rich flailing is chaotic (errors grow ~e^(50 t)); 60 steps keep f64 round-off < 1e-9
  np.testing.assert_allclose(got[:, :29], st[:, :29], rtol=0, atol=1e-9)
  eng.close()


def test_single_step_f32(torch):
  from gym_solo_amd.engine import Engine
  from oracle import solo_oracle as so
  ca, ma = make_abi('float32')
  n = 256
  eng = Engine(ca, ma, n)
  ph = so.OraclePhysics(ca, ma)
  rng = np.random.default_rng(1)
  # decorrelate the robots first
  for k in range(40):
    eng.step(torch.as_tensor(random_actions(rng, n), device='cuda', dtype=torch.float32), abi.STEP_PHYSICS)
  st = eng.state.cpu().numpy().astype(np.float64)
  a = random_actions(rng, n).astype(np.float32)
  ph.step(st, a.astype(np.float64))
  eng.step(torch.as_tensor(a, device='cuda'), abi.STEP_PHYSICS)
  got = eng.state.cpu().numpy().astype(np.float64)
  err = np.abs(got[:, :29] - st[:, :29])
  # one f32 step: positions to 1e-6, velocities (which see 1/dt-scaled motor targets) to 2e-3
  assert err[:, :15].max() < 5e-6
  assert err[:, 15:29].max() < 5e-3
  eng.close()
