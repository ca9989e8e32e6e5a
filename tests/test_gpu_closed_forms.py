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


# ---- closed forms of the CONSTRAINT rows (round 6; tests/closed_form_cases.py: the same checkers hold the CPU oracle in
#      tests/test_oracle_physics.py).  Friction, the motor clamp and the penetration push-out on the HIP engine at 4096 robots:
#      Newton's and Coulomb's laws and the model's masses are the reference here, not the oracle's step. -------------------------
def _put(eng, st):
  import torch
  eng.state.copy_(torch.as_tensor(st, device='cuda'))


def _momentum_fn(ca, ma):
  from oracle import solo_oracle as so
  ph = so.OraclePhysics(ca, ma)
  return ph, (lambda s: ph.momentum(np.ascontiguousarray(s))[0])


def test_coulomb_friction_on_the_incline_at_benchmark_scale():
  """4096 robots standing on BASELINE configs[4]'s 10-degree incline, per-robot friction as in configs[3] (set_params).
  mu > tan(theta): the centre of mass comes to rest.  mu < tan(theta): on every checked step of the slide the external impulse
  along t1 - mu n is - m g dt (sin theta - mu cos theta) to rounding, and the slide is a rigid translation with
  a = g (sin theta - mu cos theta)."""
  import torch
  import closed_form_cases as cf
  from helpers import incline_terrain
  from gym_solo_amd.model import Solo8Model
  eng, ca, ma = _engine(N, linear_damping=0.0, angular_damping=0.0, settle_steps=10)
  eng.set_terrain(incline_terrain(10.0))
  _, mom = _momentum_fn(ca, ma)
  mus = cf.incline_frictions(N, seed=3)
  eng.set_params(abi.PARAM_FRICTION, torch.as_tensor(mus, device='cuda'))
  _put(eng, cf.standing_on_incline(N))
  m = Solo8Model().total_mass
  slides = mus < np.tan(cf.THETA)
  zero = torch.zeros(N, 12, device='cuda', dtype=torch.float64)
  done, vel = 0, {}
  worst_slide = worst_stick = 0.0
  for upto in (150, 200, 275, 399):
    eng.rollout(zero.expand(upto - done, N, 12).contiguous(), abi.STEP_PHYSICS)   # one fused launch up to the checked step
    pre = eng.state.cpu().numpy().copy()
    eng.step(zero, abi.STEP_PHYSICS)                                            # the checked step: a launch of its own
    post = eng.state.cpu().numpy().copy()
    done = upto + 1
    for i in range(N):
      got, want = cf.check_coulomb_step(mom, pre[i], post[i], mus[i], ca.dt)
      if slides[i]:
        worst_slide = max(worst_slide, abs(got - want) / abs(want))
      else:
        worst_stick = max(worst_stick, abs(got))
    vel[done] = np.array([mom(post[i]) / m for i in range(N)])
  assert worst_slide < 1e-11, worst_slide
  assert worst_stick < 1e-7, worst_stick
  # "at rest": the landing's transient is still decaying (2e-6 m/s measured at worst among 2048 robots; the reference's own rest
  # test - test_solo8v2vanilla.py:77-104, 6 decimals over 10 steps - allows 1.5e-4 m/s)
  assert np.abs(vel[400][~slides]).max() < 1e-5
  a = 9.81 * (np.sin(cf.THETA) - mus[slides] * np.cos(cf.THETA))
  assert (vel[400][slides] @ cf.T1_SLOPE).max() < -0.05
  np.testing.assert_allclose(-((vel[400] - vel[201])[slides] @ cf.T1_SLOPE), a * 199 * ca.dt, rtol=1e-6)
  assert eng.stats.cpu().numpy()[5] == 0
  eng.close()


def test_saturated_motor_rows_at_benchmark_scale():
  """4096 robots at rest afloat, no gravity: M(q) du of ONE step is 0 on the base rows, + limit dt on every motor row whose
  target is out of reach (gym_solo/envs/solo8v2vanilla.py:87-90: forces = motor_torque_limit, configs.py:12), and a row that only
  holds its joint either ends at rest or sits at its bound against the motion.  M(q): the oracle's CRBA (kinematics)."""
  import torch
  import closed_form_cases as cf
  eng, ca, ma = _engine(N, gravity=(0., 0., 0.), linear_damping=0.0, angular_damping=0.0, settle_steps=1)
  ph, _ = _momentum_fn(ca, ma)
  st, acts, far, sign = cf.floating_at_rest(N)
  _put(eng, st)
  eng.step(torch.as_tensor(acts, device='cuda'), abi.STEP_PHYSICS)
  post = eng.state.cpu().numpy()
  limit_dt = ca.motor_torque_limit * ca.dt
  worst = np.zeros(3)
  held = stopped = 0
  for i in range(N):
    M = np.array(ph.step_debug(st[i].copy(), np.zeros(8)).M).reshape(abi.NV, abi.NV)
    base, sat, hold, nh, ns = cf.check_motor_clamp(M, st[i], post[i], far[i], sign[i], limit_dt)
    worst = np.maximum(worst, [base, sat, hold])
    held += nh; stopped += ns
  assert worst.max() < 1e-14, worst       # (impulse units: the bound itself is 2e-3)
  assert held > 1000 and stopped > 1000
  eng.close()


def test_penetration_push_out_at_benchmark_scale():
  """4096 robots at rest, one base sphere each 0.05 ... 2 mm inside the flat ground: after ONE step the contact point moves out
  at contact_erp d / dt, without tangential velocity."""
  import torch
  import closed_form_cases as cf
  eng, ca, ma = _engine(N, settle_steps=1)
  st, acts, d, centres, radius = cf.belly_corner_penetrating(N)
  _put(eng, st)
  eng.step(torch.as_tensor(acts, device='cuda'), abi.STEP_PHYSICS)
  post = eng.state.cpu().numpy()
  v = np.array([cf.contact_point_velocity(post[i], st[i], centres[i], radius) for i in range(N)])
  np.testing.assert_allclose(v[:, 2], ca.contact_erp * d / ca.dt, rtol=0, atol=1e-11)
  assert np.abs(v[:, :2]).max() < 1e-11
  eng.close()


@pytest.mark.parametrize('leg_mu,base_mu,slides', [(0.1, 0.5, False), (0.1, 0.1, True), (0.9, 0.1, True)])
def test_the_base_link_keeps_its_own_friction_gpu(leg_mu, base_mu, slides):
  """The reference's changeDynamics loop covers links 0 .. 11 (solo8v2vanilla.py:157-163): the base link keeps its own
  friction.  512 robots lying on their bellies on the incline, leg friction per robot through set_params as well: what decides
  is base_lateral_friction - 0.5 holds where lateral_friction = 0.1 would slide, and a slippery belly slides by Coulomb's
  closed form whatever the legs' coefficient."""
  import torch
  import closed_form_cases as cf
  from helpers import incline_terrain
  from gym_solo_amd.model import Solo8Model
  n = 512
  eng, ca, ma = _engine(n, lateral_friction=leg_mu, base_lateral_friction=base_mu, linear_damping=0.0, angular_damping=0.0, settle_steps=10)
  eng.set_terrain(incline_terrain(10.0))
  eng.set_params(abi.PARAM_FRICTION, torch.full((n,), leg_mu, device='cuda', dtype=torch.float64))   # (must not reach the belly)
  _, mom = _momentum_fn(ca, ma)
  st, acts = cf.belly_on_incline(n)
  _put(eng, st)
  a = torch.as_tensor(acts, device='cuda')
  eng.rollout(a.expand(299, n, 12).contiguous(), abi.STEP_PHYSICS)
  pre = eng.state.cpu().numpy().copy()
  eng.step(a, abi.STEP_PHYSICS)
  post = eng.state.cpu().numpy()
  m = Solo8Model().total_mass
  for i in range(0, n, 37):
    v = mom(post[i]) / m
    got, want = cf.check_coulomb_step(mom, pre[i], post[i], base_mu, ca.dt)
    if slides:
      assert v @ cf.T1_SLOPE < -0.1
      assert abs(got - want) < 1e-11 * abs(want)
    else:
      assert np.abs(v).max() < 1e-9 and abs(got) < 1e-11
  eng.close()


def test_joint_limit_rows_at_benchmark_scale():
  """4096 robots, one joint each running into its URDF limit (the reference's fixture: -10 / +10 rad): against the same step of an
  engine whose limits are far away, the row stops a joint that would cross exactly on the limit (rate C / dt), leaves one that
  would not alone, and adds an impulse to that joint's row only, pushing away from the limit.  M(q): the oracle's CRBA."""
  import torch
  import closed_form_cases as cf
  from gym_solo_amd.engine import Engine
  from gym_solo_amd.model import Solo8Model
  eng, ca, ma = _engine(N, gravity=(0., 0., 0.), linear_damping=0.0, angular_damping=0.0, motor_torque_limit=0.0, settle_steps=1)
  far = Solo8Model().to_abi()
  for j in range(abi.NUM_DOF):
    far.joint_lower[j], far.joint_upper[j] = -1e3, 1e3
  free = Engine(ca, far, N)
  ph, _ = _momentum_fn(ca, ma)
  st, dof, side, c, s = cf.joints_running_into_limits(N, margin=ca.joint_limit_margin)
  zero = torch.zeros(N, 12, device='cuda', dtype=torch.float64)
  for e in (eng, free):
    _put(e, st)
    e.step(zero, abi.STEP_PHYSICS)
  post, post_free = eng.state.cpu().numpy(), free.state.cpu().numpy()
  worst, acted = np.zeros(3), 0
  for i in range(N):
    M = np.array(ph.step_debug(st[i].copy(), np.zeros(8)).M).reshape(abi.NV, abi.NV)
    rate, off, sign, on = cf.check_joint_limit_against_free(M, st[i], post[i], post_free[i], dof[i], side[i], c[i], ca.dt)
    worst = np.maximum(worst, [rate, off, sign])
    acted += int(on)
  assert worst[0] < 1e-10 and worst[1] < 1e-13 and worst[2] == 0.0, worst
  assert 1500 < acted < 3500
  eng.close(); free.close()


def test_link_damping_of_a_pure_translation_at_benchmark_scale():
  """4096 robots translating afloat (no rotation, no joint motion, no gravity): one step leaves v0 (1 - dt k (1 + |v0|)) with
  k = linear_damping (gym_solo/core/configs.py:21 through changeDynamics, solo8v2vanilla.py:158-163) and nothing else moves."""
  import torch
  import closed_form_cases as cf
  eng, ca, ma = _engine(N, gravity=(0., 0., 0.), settle_steps=1)
  st, acts = cf.translating_afloat(N)
  _put(eng, st)
  eng.step(torch.as_tensor(acts, device='cuda'), abi.STEP_PHYSICS)
  post = eng.state.cpu().numpy()
  v0 = st[:, abi.S_LINVEL:abi.S_LINVEL + 3]
  want = v0 * (1 - ca.dt * ca.linear_damping * (1 + np.linalg.norm(v0, axis=1, keepdims=True)))
  np.testing.assert_allclose(post[:, abi.S_LINVEL:abi.S_LINVEL + 3], want, rtol=0, atol=1e-13)
  assert np.abs(post[:, abi.S_ANGVEL:abi.S_ANGVEL + 3]).max() < 1e-13 and np.abs(post[:, abi.S_QD:abi.S_QD + 8]).max() < 1e-13
  eng.close()


@pytest.mark.parametrize('leg_mu,base_mu', [(0.1, 0.5), (0.9, 0.1)])
def test_base_link_friction_in_the_f32_kernels(leg_mu, base_mu):
  """The opt-in f32 kernels (lane = row, their own assembly loop) carry the base link's coefficient per lane as well: the belly
  on the incline, 120 steps, f32 HIP engine against the f64 oracle (5e-3: f32's tolerance over a slide) - and the robot slides
  exactly when the BASE coefficient is below tan(theta)."""
  import torch
  import closed_form_cases as cf
  from helpers import incline_terrain
  from gym_solo_amd.engine import Engine
  from oracle import solo_oracle as so
  if not torch.cuda.is_available():
    pytest.fail('GPU tests need a visible MI355X')
  n = 256
  ca, ma = make_abi('float32', lateral_friction=leg_mu, base_lateral_friction=base_mu, linear_damping=0.0, angular_damping=0.0, settle_steps=10)
  terr = incline_terrain(10.0)
  eng = Engine(ca, ma, n)
  eng.set_terrain(terr)
  ph = so.OraclePhysics(ca, ma, terrain=terr)
  st, acts = cf.belly_on_incline(n)
  eng.state.copy_(torch.as_tensor(st, device='cuda', dtype=torch.float32))
  a = torch.as_tensor(acts, device='cuda', dtype=torch.float32)
  eng.rollout(a.expand(120, n, 12).contiguous(), abi.STEP_PHYSICS)
  ref = st[:2].copy()
  for _ in range(120):
    ph.step(ref, acts[:2])
  got = eng.state.cpu().numpy().astype(np.float64)
  np.testing.assert_allclose(got[:, :29], np.tile(ref[0, :29], (n, 1)), rtol=0, atol=5e-3)
  v_down = -(got[:, abi.S_LINVEL:abi.S_LINVEL + 3] @ cf.T1_SLOPE)
  assert (v_down.min() > 0.05) if base_mu < np.tan(cf.THETA) else (np.abs(v_down).max() < 1e-4)
  eng.close()
