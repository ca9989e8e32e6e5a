"""Closed forms of the physics on the HIP engine at BENCHMARK size (4096 robots, f64), independent of the oracle's step:
what a `stepSimulation` (gym_solo/envs/solo8v2vanilla.py:91) must satisfy whatever the solver does.  The oracle's own
versions of these properties (one robot) are tests/test_oracle_physics.py; here the HIP kernels are held to them directly,
in one fused launch and in single-step launches (kinematic helpers of the oracle - `momentum` - are used as the checker)."""
import numpy as np
import pytest

from gym_solo_amd import abi
from helpers import make_abi

pytestmark = pytest.mark.gpu

N = 4096


def _engine(n, **kw):
  import torch
  from gym_solo_amd.engine import Engine
  if not torch.cuda.is_available():
    pytest.fail('GPU tests need a visible MI355X')
  ca, ma = make_abi('float64', **kw)
  return Engine(ca, ma, n), ca, ma


def _afloat(st, rng, spin):
  """the settled robots lifted above the ground (they fall 0.4 m at most in these tests), random joint angles; optionally moving in every coordinate"""
  n = st.shape[0]
  st[:, abi.S_POS + 2] = rng.uniform(2.0, 3.0, n)
  st[:, abi.S_Q:abi.S_Q + 8] = rng.uniform(-1.5, 1.5, (n, 8))
  st[:, abi.S_LINVEL:abi.S_LINVEL + 3] = rng.uniform(-1.0, 1.0, (n, 3))
  if spin:
    st[:, abi.S_QD:abi.S_QD + 8] = rng.uniform(-3.0, 3.0, (n, 8))
    st[:, abi.S_ANGVEL:abi.S_ANGVEL + 3] = rng.uniform(-2.0, 2.0, (n, 3))
    q = rng.normal(size=(n, 4))
    st[:, abi.S_QUAT:abi.S_QUAT + 4] = q / np.linalg.norm(q, axis=1, keepdims=True)
  else:
    st[:, abi.S_QD:abi.S_QD + 8] = 0.0
    st[:, abi.S_ANGVEL:abi.S_ANGVEL + 3] = 0.0
  return st


@pytest.mark.parametrize('fused', [True, False])
def test_free_fall_closed_form_at_benchmark_scale(fused):
  """Semi-implicit Euler free fall, no damping, motors without torque: after k steps v = v0 + g k dt and
  p = p0 + v0 k dt + g dt^2 k (k + 1) / 2 for EVERY robot, whatever its pose; nothing else moves (all bodies fall alike)."""
  import torch
  eng, ca, _ = _engine(N, linear_damping=0.0, angular_damping=0.0, motor_torque_limit=0.0)
  rng = np.random.default_rng(5)
  st0 = _afloat(eng.state.cpu().numpy().copy(), rng, spin=False)
  eng.state.copy_(torch.as_tensor(st0, device='cuda'))
  k, dt, g = 200, ca.dt, np.array(list(ca.gravity))
  acts = torch.zeros(k, N, 12, device='cuda', dtype=torch.float64)
  if fused:
    eng.rollout(acts, abi.STEP_PHYSICS)
  else:
    for i in range(k):
      eng.step(acts[i], abi.STEP_PHYSICS)
  st = eng.state.cpu().numpy()
  v0 = st0[:, abi.S_LINVEL:abi.S_LINVEL + 3]
  np.testing.assert_allclose(st[:, abi.S_LINVEL:abi.S_LINVEL + 3], v0 + g * k * dt, rtol=0, atol=1e-12)
  np.testing.assert_allclose(st[:, abi.S_POS:abi.S_POS + 3], st0[:, abi.S_POS:abi.S_POS + 3] + v0 * k * dt + g * dt * dt * k * (k + 1) / 2,
                             rtol=0, atol=1e-12)
  np.testing.assert_allclose(st[:, abi.S_Q:abi.S_Q + 8], st0[:, abi.S_Q:abi.S_Q + 8], rtol=0, atol=1e-12)
  np.testing.assert_allclose(st[:, abi.S_QD:abi.S_QD + 8], 0.0, atol=1e-12)
  np.testing.assert_allclose(st[:, abi.S_ANGVEL:abi.S_ANGVEL + 3], 0.0, atol=1e-12)
  np.testing.assert_allclose(st[:, abi.S_QUAT:abi.S_QUAT + 4], st0[:, abi.S_QUAT:abi.S_QUAT + 4], rtol=0, atol=1e-12)
  assert eng.stats.cpu().numpy()[5] == 0
  eng.close()


def test_motor_impulses_are_internal_at_benchmark_scale():
  """No gravity, no damping, no contact: whatever the joint motors do in a step, they change neither the linear nor the
  angular momentum of their robot.  The same 4096 tumbling robots take ONE step with the motors driving towards random
  targets and one with the motors without torque: the momenta of the two new velocities, evaluated at the configuration
  the step started from, agree to rounding - while the joint rates themselves differ by rad/s."""
  import torch
  from oracle import solo_oracle as so
  rng = np.random.default_rng(9)
  on, ca, ma = _engine(N, gravity=(0., 0., 0.), linear_damping=0.0, angular_damping=0.0)
  off, _, _ = _engine(N, gravity=(0., 0., 0.), linear_damping=0.0, angular_damping=0.0, motor_torque_limit=0.0)
  st0 = _afloat(on.state.cpu().numpy().copy(), rng, spin=True)
  acts = torch.as_tensor(rng.uniform(-2 * np.pi, 2 * np.pi, (N, 12)), device='cuda')
  post = []
  for eng in (on, off):
    eng.state.copy_(torch.as_tensor(st0, device='cuda'))
    eng.step(acts, abi.STEP_PHYSICS)
    post.append(eng.state.cpu().numpy().copy())
    assert eng.stats.cpu().numpy()[5] == 0
  vel = [slice(abi.S_LINVEL, abi.S_LINVEL + 3), slice(abi.S_ANGVEL, abi.S_ANGVEL + 3), slice(abi.S_QD, abi.S_QD + 8)]
  assert np.abs(post[0][:, vel[2]] - post[1][:, vel[2]]).max() > 1.0   # the motors did act
  ph = so.OraclePhysics(ca, ma)
  worst = np.zeros(2)
  for i in range(N):
    mom = []
    for p in post:
      s = st0[i].copy()
      for sl in vel:
        s[sl] = p[i, sl]
      lin, ang, _ = ph.momentum(s)
      mom.append((lin.copy(), ang.copy()))
    worst = np.maximum(worst, [np.abs(mom[0][0] - mom[1][0]).max(), np.abs(mom[0][1] - mom[1][1]).max()])
  assert worst[0] < 1e-12 and worst[1] < 1e-12, worst
  on.close(); off.close()
