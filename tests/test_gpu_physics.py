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


@pytest.mark.parametrize('seed', range(6))
def test_random_configurations_match_oracle_f64(torch, seed):
  """Every configuration field the reference exposes (configs.py:8-38: dt, torque limit, start pose, gravity,
  damping, friction) and the engine's own knobs, at a random point of their ranges (tests/config_space.py): the settle
  loop from the tilted start pose under the tilted gravity, then 30 random-action steps - f64 engine vs oracle."""
  from gym_solo_amd.engine import Engine
  from oracle import solo_oracle as so
  from config_space import random_config
  kw = random_config(seed)
  ca, ma = make_abi('float64', **kw)
  n = 32
  eng = Engine(ca, ma, n)
  ph = so.OraclePhysics(ca, ma)
  ref = ph.settle(1)
  np.testing.assert_allclose(eng.snapshot.cpu().numpy()[:, :29], np.tile(ref[:, :29], (n, 1)), rtol=0, atol=1e-9, err_msg=str(kw))
  st = eng.state.cpu().numpy().copy()
  rng = np.random.default_rng(100 + seed)
  for k in range(30):
    a = random_actions(rng, n)
    ph.step(st, a)
    eng.step(torch.as_tensor(a, device='cuda'), abi.STEP_PHYSICS)
  np.testing.assert_allclose(eng.state.cpu().numpy()[:, :29], st[:, :29], rtol=0, atol=1e-9, err_msg=str(kw))
  eng.close()


@pytest.mark.parametrize('seed', [11, 12, 13])
def test_random_configuration_terrain_and_parameters_together_f64(torch, seed):
  """The extensions TOGETHER with a non-default configuration: a random point of the configuration space
  (tests/config_space.py), a bumpy heightfield and per-robot friction / base-mass scale (BASELINE configs[3] + [4] on
  one engine) - settle on the terrain with the randomised robots, then 25 random-action steps, f64 engine vs oracle."""
  import helpers
  from gym_solo_amd.engine import Engine
  from oracle import solo_oracle as so
  from config_space import random_config
  kw = random_config(seed)
  ca, ma = make_abi('float64', **kw)
  n = 16
  rng = np.random.default_rng(200 + seed)
  terrain = helpers.bumpy_terrain(seed=seed)
  params = np.zeros((n, 4))
  params[:, 0] = rng.uniform(0.3, 1.0, n)
  params[:, 1] = rng.uniform(0.8, 1.2, n)
  eng = Engine(ca, ma, n)
  eng.set_params(abi.PARAM_FRICTION, torch.as_tensor(params[:, 0], device='cuda').contiguous())
  eng.set_params(abi.PARAM_BASE_MASS_SCALE, torch.as_tensor(params[:, 1], device='cuda').contiguous())
  eng.set_terrain(terrain)   # (re-runs the settle loop: on the new ground, with the per-robot parameters)
  ph = so.OraclePhysics(ca, ma, terrain=terrain)
  st = ph.settle(n, params=params)
  np.testing.assert_allclose(eng.snapshot.cpu().numpy()[:, :29], st[:, :29], rtol=0, atol=1e-8, err_msg=str(kw))
  st = eng.state.cpu().numpy().copy()
  for k in range(25):
    a = random_actions(rng, n)
    ph.step(st, a, params=params)
    eng.step(torch.as_tensor(a, device='cuda'), abi.STEP_PHYSICS)
  np.testing.assert_allclose(eng.state.cpu().numpy()[:, :29], st[:, :29], rtol=0, atol=3e-8, err_msg=str(kw))
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


@pytest.mark.parametrize('seed', range(8))
def test_single_step_f32_over_random_configurations(torch, seed):
  """The f32 kernel - the throughput headline - at random points of the configuration space (tests/config_space.py):
  one step of the f32 engine against the f64 oracle from the engine's own state after 40 decorrelating steps.
  Measured over the 8 points: positions / quaternion / joint angles 1.2e-7 ... 2.8e-7, velocities 2e-5 ... 1.5e-4
  (they see the solver's round-off scaled by 1 / dt); the bounds are ~10x that."""
  from gym_solo_amd.engine import Engine
  from oracle import solo_oracle as so
  from config_space import random_config
  kw = random_config(seed)
  ca, ma = make_abi('float32', **kw)
  ca64, _ = make_abi('float64', **kw)
  n = 256
  eng = Engine(ca, ma, n)
  ph = so.OraclePhysics(ca64, ma)
  rng = np.random.default_rng(seed)
  for k in range(40):
    eng.step(torch.as_tensor(random_actions(rng, n), device='cuda', dtype=torch.float32), abi.STEP_PHYSICS)
  st = eng.state.cpu().numpy().astype(np.float64)
  a = random_actions(rng, n).astype(np.float32)
  ph.step(st, a.astype(np.float64))
  eng.step(torch.as_tensor(a, device='cuda'), abi.STEP_PHYSICS)
  got = eng.state.cpu().numpy().astype(np.float64)
  assert np.isfinite(got[:, :29]).all()
  err = np.abs(got[:, :29] - st[:, :29])
  assert err[:, :15].max() < 3e-6, (kw, err[:, :15].max())
  assert err[:, 15:29].max() < 2e-3, (kw, err[:, 15:29].max())
  eng.close()


def _sway_actions(k, n):
  """Smooth stand-up and sway (contact-rich, not chaotic: a 1e-9 perturbation of the start state
  stays below 1e-6 over the 1000 steps on the f64 oracle; larger / faster sways turn into a
  stick-slip gait that amplifies round-off 1e5-fold - measured while choosing these numbers)."""
  t = k * 1e-3
  a = np.zeros((n, 12))
  amp = 0.2 * min(1.0, t / 0.3)
  for leg in range(4):
    s = 1.0 if leg < 2 else -1.0
    a[:, 3 * leg] = s * (0.5 + amp * np.sin(2 * np.pi * 1.0 * t + leg))
    a[:, 3 * leg + 1] = -s * (1.0 + amp * np.sin(2 * np.pi * 1.0 * t + leg))
  return a


@pytest.mark.parametrize('regime', ['rest', 'stand-sway'])
def test_thousand_step_divergence_within_1e4(torch, regime):
  """BASELINE target: <= 1e-4 relative joint-state divergence over 1000 steps.  It is checked
  against the f64 CPU oracle (PyBullet is unavailable) in the two non-chaotic regimes: `rest`
  (zero targets from the folded reset pose: the robot unfolds and stands at z = 0.337, the
  reference's standing height) and a smooth contact-rich stand-and-sway.  The f32 engine stays
  within 1e-4 on q and qd (base pose within 1e-3 of a metre), the f64 engine within 1e-9 (1e-8 on
  the joint rates, which see 1/dt-scaled motor targets; measured 6e-11 / 1.2e-9 after 1000 steps).
  (Random U(-2pi, 2pi) flailing is chaotic: there round-off grows ~e^(50 t) in ANY arithmetic —
  measured in tests/measure_divergence.py, quoted in DESIGN.md.)"""
  from gym_solo_amd.engine import Engine
  from oracle import solo_oracle as so
  n = 8
  ca32, ma = make_abi('float32')
  ca64, _ = make_abi('float64')
  e32, e64 = Engine(ca32, ma, n), Engine(ca64, ma, n)
  ph = so.OraclePhysics(ca64, ma)
  st = e64.state.cpu().numpy().copy()
  for k in range(1000):
    a = np.zeros((n, 12)) if regime == 'rest' else _sway_actions(k, n)
    ph.step(st, a)
    e64.step(torch.as_tensor(a, device='cuda'), abi.STEP_PHYSICS)
    e32.step(torch.as_tensor(a, device='cuda', dtype=torch.float32), abi.STEP_PHYSICS)
  s64, s32 = e64.state.cpu().numpy(), e32.state.cpu().numpy().astype(np.float64)

  def rel(x, sl):
    return (np.abs(x[:, sl] - st[:, sl]).max(axis=1) / np.maximum(np.abs(st[:, sl]).max(axis=1), 1.0)).max()
  q, qd, base = slice(7, 15), slice(21, 29), slice(0, 7)
  assert rel(s64, q) < 1e-9 and rel(s64, qd) < 1e-8 and rel(s64, base) < 1e-9
  # f32: positions well inside the 1e-4 bar (measured 2e-7 / 1e-7 at 1000 steps); the momentary joint
  # RATES carry the solver's f32 round-off scaled by 1/dt (motor rows target kp (q* - q) / dt): 2e-7 at
  # rest, 1e-4 ... 1e-3 while the legs move (tests/measure_divergence.py), hence the 1e-3 bound
  assert rel(s32, q) < 1e-4 and rel(s32, qd) < (1e-4 if regime == 'rest' else 1e-3) and rel(s32, base) < 1e-3
  assert 0.25 < np.median(st[:, 2]) < 0.36   # standing, not lying
  if regime == 'rest':
    assert abs(np.median(st[:, 2]) - 0.33698) < 2e-3  # examples/solo8_vanilla/interactive_pos_control.py:23
  e32.close()
  e64.close()


def test_passive_rest_pose_reproduces_the_reference_vector_gpu():
  """The reference's only pybullet-extracted state (test_obs_observations.py:256-275) on the f64 HIP
  engine: weak motors + the vector's sign pattern -> the recorded |HFE| = 1.53013, |KFE| = 3.08532
  (see tests/test_oracle_physics.py for the reasoning), and the engine agrees with the oracle."""
  import torch
  from gym_solo_amd.engine import Engine
  from helpers import make_abi
  from oracle import solo_oracle as so
  from test_oracle_physics import reference_rest_case
  if not torch.cuda.is_available():
    pytest.fail('GPU tests need a visible MI355X')
  q_ref, _, targets = reference_rest_case()
  ca, ma = make_abi('float64', motor_torque_limit=0.02, settle_steps=3000, starting_joint_pos=targets)
  eng = Engine(ca, ma, 4)
  snap = eng.snapshot.cpu().numpy()
  q = snap[:, abi.S_Q:abi.S_Q + 8]
  np.testing.assert_allclose(q, np.tile(q_ref[[0, 1, 3, 4, 6, 7, 9, 10]], (4, 1)), rtol=0, atol=5e-4)
  assert np.abs(snap[:, abi.S_QD:abi.S_QD + 8]).max() < 1e-9
  home = so.OraclePhysics(ca, ma).settle(1)
  np.testing.assert_allclose(snap[:, :29], np.tile(home[:, :29], (4, 1)), rtol=0, atol=1e-8)
  eng.close()


@pytest.mark.parametrize('resid', [0.0, 1e-7])
@pytest.mark.parametrize('dtype,n,spl,streams,terrain', [('float32', 4096, 20, 1, None), ('float32', 8192, 20, 1, None), ('float64', 4096, 20, 1, None),
                                                         ('float32', 4096, 10, 2, 'stairs'), ('float32', 4096, 1, 1, None), ('float64', 4096, 5, 2, 'incline')])
def test_identical_robots_stay_identical(torch, dtype, n, spl, streams, terrain, resid):
  """Every robot of a batch starts in the same state and receives the same actions: whatever the number of
  waves per SIMD, all of them must end in the SAME bits, and in the bits of a 64-robot batch (one wave per
  SIMD).  A wave that reads anything of another wave's - registers or LDS beyond its allocation, a stale scalar -
  shows up here as robots that differ; this is the test that caught the round-3 assembly loop computing
  wave-dependent garbage at 2+ waves per SIMD.  20 steps with observations, rewards and
  terminations - fused, in stream slices, one launch per step (in-place outputs), on heightfields -, the settle loop
  (physics-only kernel) before them; default solver and pybullet's residual threshold (its own kernels)."""
  import helpers
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig, Solo8VanillaEnv
  from gym_solo_amd.workloads import register_benchmark_workload
  tdt = torch.float32 if dtype == 'float32' else torch.float64
  g = torch.Generator(device='cuda').manual_seed(8)
  one = (torch.rand(20, 1, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * (2 * np.pi)
  ref = None
  for count in (64, n):
    cfg = Solo8VanillaConfig()
    cfg.dtype, cfg.num_envs, cfg.auto_reset, cfg.steps_per_launch, cfg.rollout_streams = dtype, count, True, spl, streams
    cfg.solver_residual_threshold = resid
    if terrain is not None:
      cfg.terrain = getattr(helpers, terrain + '_terrain')()
    env = Solo8VanillaEnv(config=cfg, copy_outputs=False)
    register_benchmark_workload(env, max_steps=13)   # an episode end (and restore) inside the launch
    env._ensure_program()
    eng = env.engine
    snap = eng.snapshot.cpu().numpy()
    assert (snap == snap[0]).all(), 'the settle loop left different robots'
    out = eng.rollout(one.expand(20, count, 12).contiguous(), abi.STEP_ALL, record=True)
    eng.synchronize()
    got = [eng.state.cpu().numpy(), eng.cost.cpu().numpy()] + [t.cpu().numpy() for t in out]
    assert all((x == x[0:1]).all() for x in got[:2]), 'robots of one batch differ'
    assert all((x == x[:, 0:1]).all() for x in got[2:]), 'recorded outputs of one batch differ'
    if ref is None:
      ref = [got[0][0], got[1][0]] + [x[:, 0] for x in got[2:]]
    else:
      for a, b in zip(ref, [got[0][0], got[1][0]] + [x[:, 0] for x in got[2:]]):
        np.testing.assert_array_equal(a, b)
    env._close()


def test_convergence_tolerance_changes_sweeps_not_physics():
  """SoloConfig.solver_ulp_tolerance (f64 default 512 half-ulps since round 6; 2 before; 0 = exact fixed point): whatever the
  tolerance, 60 contact-rich flailing steps stay within 1e-9 of the oracle - whose solver has no convergence test at all and runs
  its 50 plain sweeps - while the sweeps a robot-step takes go down; and the f32 default stays at 2 (512 f32 half-ulps would be
  3e-5 relative)."""
  import torch
  from gym_solo_amd.engine import Engine
  from oracle import solo_oracle as so
  from helpers import make_abi, random_actions
  if not torch.cuda.is_available():
    pytest.fail('GPU tests need a visible MI355X')
  assert make_abi('float64')[0].solver_ulp_tolerance == 512 and make_abi('float32')[0].solver_ulp_tolerance == 2
  n = 512
  rng = np.random.default_rng(7)
  a = np.stack([random_actions(rng, n) for _ in range(60)])
  ca, ma = make_abi('float64')
  ref = None
  sweeps = {}
  for tol in (0, 2, 512):
    ca, ma = make_abi('float64', steps_per_launch=60, solver_ulp_tolerance=tol)
    eng = Engine(ca, ma, n)
    if ref is None:
      ph = so.OraclePhysics(ca, ma)
      ref = eng.state.cpu().numpy().copy()
      for k in range(60):
        ph.step(ref, a[k], threads=8)
    eng.rollout(torch.as_tensor(a, device='cuda'), abi.STEP_PHYSICS)
    err = np.abs(eng.state.cpu().numpy()[:, :29] - ref[:, :29]).max()
    sweeps[tol] = float(eng.cost.double().mean())
    assert err < 1e-9, (tol, err)
    eng.close()
  assert sweeps[0] >= sweeps[2] > sweeps[512]
