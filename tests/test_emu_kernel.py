"""The PRODUCT kernel source on the CPU fibre emulator (tests/emu): physics parity with the oracle,
fused multi-step launches, and an UndefinedBehaviorSanitizer build (the GPU pool cannot run
sanitizers)."""
import numpy as np
import pytest

from emu_kernel import EmuEngine
from gym_solo_amd import abi
from helpers import make_abi, random_actions
from oracle import solo_oracle as so


def bench_program():
  """The benchmark's SoloProgram, compiled by the host factories from an engine-less env."""
  from test_env_host import make_env
  from gym_solo_amd.workloads import register_benchmark_workload
  env = make_env()
  register_benchmark_workload(env, max_steps=4)
  env._ensure_program()
  return env.engine.program


@pytest.mark.parametrize('dtype,tol', [('float64', 1e-10), ('float32', 2e-3)])
def test_physics_matches_oracle(dtype, tol):
  ca, ma = make_abi(dtype)
  ph = so.OraclePhysics(ca, ma)
  n = 3
  st = np.tile(ph.settle(1), (n, 1))
  e = EmuEngine(ca, ma, n)
  e.state[:] = st
  rng = np.random.default_rng(0)
  for k in range(25):
    a = random_actions(rng, n)
    ph.step(st, a)
    e.step(a, abi.STEP_PHYSICS)
  np.testing.assert_allclose(e.state[:, :29], st[:, :29], rtol=0, atol=tol)


@pytest.mark.parametrize('seed', range(4))
def test_random_configurations_match_oracle(seed):
  """Every configuration field the reference exposes (configs.py:8-38: dt, torque limit, start pose, gravity,
  damping, friction) and the engine's own knobs at a random point of their ranges (tests/config_space.py): the
  kernel source on the emulator against the oracle, from the oracle's post-settle state, 25 random-action steps
  (the GPU twin of this test also runs the settle loop from the tilted start pose: tests/test_gpu_physics.py)."""
  from config_space import random_config
  kw = random_config(seed)
  ca, ma = make_abi('float64', **kw)
  ph = so.OraclePhysics(ca, ma)
  n = 2
  st = np.tile(ph.settle(1), (n, 1))
  e = EmuEngine(ca, ma, n)
  e.state[:] = st
  rng = np.random.default_rng(100 + seed)
  for k in range(25):
    a = random_actions(rng, n)
    ph.step(st, a)
    e.step(a, abi.STEP_PHYSICS)
  np.testing.assert_allclose(e.state[:, :29], st[:, :29], rtol=0, atol=1e-10, err_msg=str(kw))


def test_fused_rollout_equals_single_steps():
  """steps_per_launch > 1 is the same arithmetic: bit-identical trajectories and outputs,
  including an auto-reset in the middle of the fused launch."""
  ca, ma = make_abi('float32', auto_reset=True, settle_steps=60)
  prog = bench_program()
  rng = np.random.default_rng(1)
  acts = rng.uniform(-6, 6, (9, 2, 12))
  a = EmuEngine(ca, ma, 2, program=prog)
  a.settle()
  b = EmuEngine(ca, ma, 2, program=prog)
  b.state[:] = a.state
  b.snapshot[:] = a.snapshot
  obs, rew, done = [], [], []
  for k in range(9):
    a.step(acts[k])
    obs.append(a.obs.copy()); rew.append(a.reward.copy()); done.append(a.done.copy())
  fo, fr, fd = b.rollout(acts)
  np.testing.assert_array_equal(np.stack(obs), fo)
  np.testing.assert_array_equal(np.stack(rew), fr)
  np.testing.assert_array_equal(np.stack(done), fd)
  np.testing.assert_array_equal(a.state, b.state)
  np.testing.assert_array_equal(a.term_count, b.term_count)
  np.testing.assert_array_equal(a.stats, b.stats)
  assert fd[4].all() and fd.sum() == 2  # TimeBased(4) fires on the 5th step of each robot


@pytest.mark.parametrize('dtype,n,k,chunk', [('float32', 2, 9, 2), ('float64', 3, 9, 4), ('float64', 16, 7, 3), ('float32', 8, 40, 32)])
def test_robot_migration_is_scheduling_only(dtype, n, k, chunk):
  """SoloConfig.migrate_steps: a fused launch cut into chunks whose robots travel from wave to wave through the
  launch's work queue (one ring for 2 / 3 robots, eight rings for 8 / 16; ragged last chunks; more steps than one
  epilogue pass holds) - the same trajectories, outputs, counters, statistics and per-robot sweep counts, bit for bit,
  including an auto-reset inside the launch."""
  prog = bench_program()
  rng = np.random.default_rng(7)
  acts = rng.uniform(-6, 6, (k, n, 12))
  out = []
  for migrate in (0, chunk):
    ca, ma = make_abi(dtype, auto_reset=True, settle_steps=40, migrate_steps=migrate)
    e = EmuEngine(ca, ma, n, program=prog)
    if not out:
      e.settle()
      start = (e.state.copy(), e.snapshot.copy())
    e.state[:], e.snapshot[:] = start
    o, r, d = e.rollout(acts)
    out.append((o, r, d, e.state.copy(), e.term_count.copy(), e.stats.copy(), e.cost.copy(), e.targets.copy()))
  for a, b in zip(*out):
    np.testing.assert_array_equal(a, b)
  assert out[1][2][4].all()  # TimeBased(4): every robot ends its first episode on the fifth step


def test_a_wave_that_gives_up_waiting_says_so():
  """The migration queue's waits are bounded (a launch must not hang on a bug) - and a wave that gives up must not pass
  for success: it counts itself in slot 6 of the statistics and sets the engine's fault word, which the
  C-ABI turns into SOLO_ERR_INCOMPLETE on every later call.  FAULT INJECTION on the emulator: ring 0's tail starts one
  slot too far, so the first chunk-1 slot is never published; the wave holding that ticket gives up, every other ticket
  is served, and exactly one robot of the launch is left half stepped."""
  prog = bench_program()
  n, k, chunk = 4, 8, 4
  rng = np.random.default_rng(3)
  acts = rng.uniform(-6, 6, (k, n, 12))
  ca, ma = make_abi('float64', auto_reset=True, settle_steps=40, migrate_steps=chunk)
  e = EmuEngine(ca, ma, n, program=prog)
  e.settle()
  start = (e.state.copy(), e.snapshot.copy())
  e.rollout(acts)
  good = e.state.copy()
  assert e.lib.solo_emu_take_fault() == 0 and e.stats[:, 6].sum() == 0
  e.state[:], e.snapshot[:] = start
  e.term_count[:] = 0
  e.lib.solo_emu_sabotage_queue(1)
  try:
    e.rollout(acts)
  finally:
    e.lib.solo_emu_sabotage_queue(0)
  assert e.lib.solo_emu_take_fault() == 1 and e.stats[:, 6].sum() == 1
  left_behind = [r for r in range(n) if not np.array_equal(e.state[r], good[r])]
  assert len(left_behind) == 1, left_behind


def test_ubsan_build_runs_clean():
  """-fsanitize=undefined,bounds-strict build of the kernel source: aborts on the first report."""
  ca, ma = make_abi('float32', settle_steps=30)
  prog = bench_program()
  e = EmuEngine(ca, ma, 2, program=prog, variant='ubsan')
  e.settle()
  rng = np.random.default_rng(2)
  for k in range(6):
    e.step(rng.uniform(-6, 6, (2, 12)))
  assert np.isfinite(e.state).all() and np.isfinite(e.obs).all()


@pytest.mark.parametrize('kind', ['incline', 'stairs', 'bumpy'])
def test_heightfield_terrain_matches_oracle(kind):
  """BASELINE configs[4]: sphere vs heightfield tangent plane (bilinear height + normal lookup per
  collision sphere); emulated kernel vs the oracle, robots dropped onto the terrain."""
  import helpers
  terrain = getattr(helpers, kind + '_terrain')()
  ca, ma = make_abi('float64', settle_steps=0)
  ph = so.OraclePhysics(ca, ma, terrain=terrain)
  n = 2
  st = ph.initial_state(n)
  st[:, abi.S_POS + 2] = 0.12  # folded legs, just above the ground: contacts within a few steps
  st[:, abi.S_Q:abi.S_Q + 8] = [np.pi / 2, np.pi, np.pi / 2, np.pi, -np.pi / 2, -np.pi, -np.pi / 2, -np.pi]
  st[1, abi.S_POS] = 0.4
  st[1, abi.S_POS + 2] = 0.25
  st[1, abi.S_QUAT:abi.S_QUAT + 4] = [0.1, -0.2, 0.3, 0.9273618495495703]
  e = EmuEngine(ca, ma, n, terrain=terrain)
  e.state[:] = st
  rng = np.random.default_rng(3)
  touched = 0
  for k in range(150):
    a = np.array([np.pi / 2, np.pi, 0, np.pi / 2, np.pi, 0, -np.pi / 2, -np.pi, 0, -np.pi / 2, -np.pi, 0]) + rng.uniform(-1, 1, (n, 12))
    ph.step(st, a)
    e.step(a, abi.STEP_PHYSICS)
  np.testing.assert_allclose(e.state[:, :29], st[:, :29], rtol=0, atol=1e-9)
  # the robots really are resting on / colliding with the terrain, not falling through it
  assert st[:, abi.S_POS + 2].min() > -0.2 and np.abs(st[:, abi.S_LINVEL + 2]).max() < 3.0


def test_zero_heightfield_equals_flat_plane():
  from gym_solo_amd import abi as _abi
  ca, ma = make_abi('float64', settle_steps=0)
  flat = _abi.make_terrain(np.zeros((8, 8)), 0.5)
  a, b = EmuEngine(ca, ma, 1), EmuEngine(ca, ma, 1, terrain=flat)
  for e in (a, b):
    e.state[0, abi.S_POS + 2] = 0.1
  rng = np.random.default_rng(5)
  for k in range(40):
    act = rng.uniform(-2, 2, (1, 12))
    a.step(act, abi.STEP_PHYSICS)
    b.step(act, abi.STEP_PHYSICS)
  np.testing.assert_allclose(a.state, b.state, rtol=0, atol=1e-12)


@pytest.mark.parametrize('dtype,tol', [('float64', 1e-12), ('float32', 2e-5)])
def test_ulp_tolerance_only_moves_last_bits(dtype, tol):
  """SoloConfig.solver_ulp_tolerance: 0 ends the Gauss-Seidel iteration only at a bit-exact
  fixed point; the default 2 also ends last-bit limit cycles.  One step from identical states
  differs by rounding noise only (positions/angles to `tol`; velocities carry the 1/dt-scaled
  motor rows, hence x1e3)."""
  ca0, ma = make_abi(dtype, solver_ulp_tolerance=0)
  ca2, _ = make_abi(dtype, solver_ulp_tolerance=2)
  assert ca0.solver_ulp_tolerance == 0 and ca2.solver_ulp_tolerance == 2
  n = 6
  a = EmuEngine(ca0, ma, n)
  rng = np.random.default_rng(5)
  for k in range(60):  # flail into contact-rich, decorrelated poses (exact mode)
    a.step(random_actions(rng, n), abi.STEP_PHYSICS)
  b = EmuEngine(ca2, ma, n)
  worst_q = worst_v = 0.0
  for k in range(10):
    b.state[:] = a.state
    act = random_actions(rng, n)
    a.step(act, abi.STEP_PHYSICS)
    b.step(act, abi.STEP_PHYSICS)
    d = np.abs(a.state - b.state)
    worst_q = max(worst_q, d[:, :15].max())
    worst_v = max(worst_v, d[:, 15:29].max())
  assert worst_q < tol and worst_v < 1e3 * tol, (worst_q, worst_v)


def test_negative_ulp_tolerance_rejected():
  with pytest.raises(ValueError):
    make_abi('float32', solver_ulp_tolerance=-1)


def test_flailing_batch_matches_oracle_every_step():
  """A wider batch of flailing robots on the incline, checked after EVERY step: slowly converging
  row pairs (a saturating motor against a base-corner contact) make the sparse Gauss-Seidel
  revisit the same few rows sweep after sweep, which is where a stale row-vector buffer in the
  double-buffered fetch showed up (one robot, one step, 8e-4) before it was fixed."""
  import helpers
  terrain = helpers.incline_terrain()
  ca, ma = make_abi('float64')
  n = 16
  ph = so.OraclePhysics(ca, ma, terrain=terrain)
  st = np.tile(ph.settle(1), (n, 1))
  e = EmuEngine(ca, ma, n, terrain=terrain)
  e.state[:] = st
  rng = np.random.default_rng(9)
  for k in range(40):
    a = random_actions(rng, n)
    ph.step(st, a)
    e.step(a, abi.STEP_PHYSICS)
    np.testing.assert_allclose(e.state[:, :29], st[:, :29], rtol=0, atol=1e-9, err_msg='step %d' % k)


def test_all_sixteen_spheres_in_contact():
  """More than 12 touching spheres (impossible on a plane, possible on a heightfield): a robot
  wedged into a trench narrower than its base has all 16 collision spheres in contact (56
  constraint rows).  The solver has no contact cap; kernel vs oracle over the violent first steps."""
  import helpers
  terrain = helpers.trench_terrain()
  ca, ma = make_abi('float64', settle_steps=0)
  ph = so.OraclePhysics(ca, ma, terrain=terrain)
  st = ph.initial_state(1)
  st[:, abi.S_POS + 2] = 0.08
  st[:, abi.S_Q:abi.S_Q + 8] = [np.pi / 2, np.pi, np.pi / 2, np.pi, -np.pi / 2, -np.pi, -np.pi / 2, -np.pi]
  a = np.array([[np.pi / 2, np.pi, 0, np.pi / 2, np.pi, 0, -np.pi / 2, -np.pi, 0, -np.pi / 2, -np.pi, 0]])
  e = EmuEngine(ca, ma, 1, terrain=terrain)
  e.state[:] = st
  contacts = []
  for k in range(4):
    dbg = ph.step_debug(st[0].copy(), a[0][[0, 1, 3, 4, 6, 7, 9, 10]])
    contacts.append((dbg.num_rows - 8) // 3)
    ph.step(st, a)
    e.step(a, abi.STEP_PHYSICS)
    np.testing.assert_allclose(e.state[:, :29], st[:, :29], rtol=1e-9, atol=1e-9, err_msg='step %d' % k)
  assert max(contacts) == 16


@pytest.mark.parametrize('resid,on_limit', [(0.0, 1e-9), (1e-7, 1e-6)])
@pytest.mark.parametrize('dtype,tol', [('float64', 1e-10), ('float32', 2e-3)])
def test_joint_limit_rows_match_oracle(dtype, tol, resid, on_limit):
  """URDF joint limits as unilateral rows on lanes k = 14, 15 ([recalled]
  btMultiBodyJointLimitConstraint): joints driven into +-10 rad stop there, emulator == oracle.  Solved to the
  fixed point (solver_residual_threshold 0) the joints sit ON the limit to 1e-9 rad; with pybullet's default
  residual threshold (1e-7: a velocity residual of up to 3.2e-4 rad/s per row is accepted) to 1e-6."""
  from helpers import joint_limit_case
  ca, ma = make_abi(dtype, solver_residual_threshold=resid)
  ph = so.OraclePhysics(ca, ma)
  st, tg = joint_limit_case(ph, n=3)
  e = EmuEngine(ca, ma, 3)
  e.state[:] = st
  hit = np.zeros(st.shape[0], dtype=bool)
  for k in range(25):
    ph.step(st, tg)
    e.step(tg, abi.STEP_PHYSICS)
    assert np.abs(st[:, abi.S_Q:abi.S_Q + 8]).max() <= 10.0 + on_limit
    hit |= (np.abs(st[:, abi.S_Q:abi.S_Q + 8]) > 10.0 - on_limit).any(axis=1)
  assert hit.all()   # every robot has a joint sitting ON its limit
  np.testing.assert_allclose(e.state[:, :29], st[:, :29], rtol=0, atol=tol)


@pytest.mark.parametrize('dtype,tol', [('float64', 1e-10), ('float32', 2e-3)])
def test_residual_threshold_exit_matches_oracle(dtype, tol):
  """pybullet's solverResidualThreshold (SoloConfig::solver_residual_threshold, off by default): with 1e-7 the
  iteration ends after the first sweep whose largest squared velocity-level change (delta impulse x A_rr)^2 is below
  it - emulated kernel == oracle on contact-rich random steps, far fewer sweeps than the fixed-point iteration, and a
  result within the residual of it; a huge threshold is exactly ONE sweep."""
  n = 6
  rng = np.random.default_rng(12)
  acts = [random_actions(rng, n) for _ in range(12)]
  out = {}
  for thr in (0.0, 1e-7, 1e9):
    ca, ma = make_abi(dtype, settle_steps=0, solver_residual_threshold=thr)
    ph = so.OraclePhysics(ca, ma)
    e = EmuEngine(ca, ma, n)
    ca_settle, _ = make_abi('float64')
    st = np.tile(so.OraclePhysics(ca_settle, ma).settle(1), (n, 1))   # at rest on the ground: knees, feet, belly in contact
    e.state[:] = st
    sweeps = 0
    for a in acts:
      ph.step(st, a)
      e.step(a, abi.STEP_PHYSICS)
      sweeps += int(e.cost.sum())
    np.testing.assert_allclose(e.state[:, :29], st[:, :29], rtol=0, atol=tol, err_msg='threshold %g' % thr)
    out[thr] = (e.state.copy(), sweeps)
  assert out[1e9][1] == n * len(acts)              # one sweep per robot-step
  assert out[1e-7][1] < 0.6 * out[0.0][1]          # far fewer sweeps than the fixed-point iteration ...
  assert np.abs(out[1e-7][0][:, :29] - out[0.0][0][:, :29]).max() < 5e-2   # ... for a nearby result


@pytest.mark.parametrize('dtype,tol', [('float64', 1e-10), ('float32', 2e-3)])
@pytest.mark.parametrize('factor', [1.0, 0.85])
def test_warm_start_matches_oracle(dtype, tol, factor):
  """SoloConfig.solver_warm_start (an opt-in of the residual-threshold solver): every step's iteration starts from
  factor x the impulses the previous step ended with, clamped to the new bounds - emulated kernel == oracle (whose
  dense Gauss-Seidel starts from the same point) over contact-rich random steps with the cache carried from step to
  step; the cache itself agrees; fewer sweeps than the cold start; a reset empties the cache."""
  n = 5
  rng = np.random.default_rng(21)
  acts = [random_actions(rng, n, scale=1.5) for _ in range(14)]
  sweeps = {}
  for warm in (0.0, factor):
    ca, ma = make_abi(dtype, solver_residual_threshold=1e-7, solver_warm_start=warm, settle_steps=450)
    ph = so.OraclePhysics(ca, ma)
    st = ph.settle(n)    # (on the ground, folded: every knee and foot in contact)
    cache = np.zeros((n, 64))
    e = EmuEngine(ca, ma, n)
    e.state[:] = st
    total = 0
    for a in acts:
      ph.step(st, a, warm=cache if warm else None)
      e.step(a, abi.STEP_PHYSICS)
      total += int(e.cost.sum())
      np.testing.assert_allclose(e.state[:, :29], st[:, :29], rtol=0, atol=tol)
      if warm and dtype == 'float64':
        np.testing.assert_allclose(e.warm, cache, rtol=0, atol=1e-9)
    sweeps[warm] = total
    if warm:
      assert np.abs(e.warm).max() > 0
  assert sweeps[factor] < sweeps[0.0]
  with pytest.raises(ValueError):
    make_abi(dtype, solver_warm_start=1.0)          # needs the residual threshold
  with pytest.raises(ValueError):
    make_abi(dtype, solver_residual_threshold=1e-7, solver_warm_start=1.5)


def test_warm_start_does_not_rescue_the_residual_threshold_at_rest():
  """The objection to pybullet's residual threshold as the default (core/configs.py) is a resting robot: the reference's
  one recorded state rests at 1e-11 rad/s.  Measured on the f64 oracle (tools/rest_drift_probe.py,
  profiles/round4_rest_drift.log), a robot standing under zero targets for 3000 steps: iterated to the fixed point its
  joint rates fall to 5e-9 (and keep falling); with the threshold 1e-7 they stay at 8e-7 cold, 5e-7 with a warm start
  of 0.85 - and GROW to 1.6e-4 with a warm start of 1.0, which carries the accepted residual from step to step.  The
  warm start speeds the threshold solver up; it does not make it rest: the default stays the fixed-point iteration."""
  base = so.OraclePhysics(*make_abi('float64')).settle(1)
  rate = {}
  for thr, warm in ((1e-20, 0.0), (1e-7, 0.0), (1e-7, 0.85), (1e-7, 1.0)):
    ca, ma = make_abi('float64', solver_residual_threshold=thr, solver_warm_start=warm)
    ph = so.OraclePhysics(ca, ma)
    st, cache, zero = base.copy(), np.zeros((1, 64)), np.zeros((1, 12))
    for k in range(3000):
      ph.step(st, zero, warm=cache if warm else None)
    rate[(thr, warm)] = np.abs(st[0, abi.S_QD:abi.S_QD + 8]).max()
    assert 0.33 < st[0, 2] < 0.34
  assert rate[(1e-20, 0.0)] < 2e-8
  assert rate[(1e-7, 0.0)] > 10 * rate[(1e-20, 0.0)]
  assert rate[(1e-7, 0.85)] > 10 * rate[(1e-20, 0.0)]
  assert rate[(1e-7, 1.0)] > 10 * rate[(1e-7, 0.0)]


def test_workgroup_to_robot_map_is_a_bijection_with_contiguous_ranges_per_xcd():
  """xcd_contiguous (solo_kernel_params.h): workgroups b, b + 8, b + 16 ... - the ones that share an XCD - step a
  contiguous range of robots, and every robot of the launch is stepped exactly once, for any launch size."""
  import emu_kernel
  lib = emu_kernel.load()
  for count in (1, 2, 7, 8, 9, 13, 64, 100, 2048, 4096, 4097):
    got = [lib.solo_emu_xcd_contiguous(b, count) for b in range(count)]
    assert sorted(got) == list(range(count)), count
    for x in range(min(8, count)):
      mine = got[x::8]
      assert mine == list(range(mine[0], mine[0] + len(mine))), (count, x)


# ---- closed forms of the constraint rows on the PRODUCT KERNEL SOURCE (CPU emulator; tests/closed_form_cases.py - the GPU runs
#      them at 4096 robots in tests/test_gpu_closed_forms.py, the oracle in tests/test_oracle_physics.py) -----------------------
def _kin(ca, ma):
  from oracle import solo_oracle as so
  ph = so.OraclePhysics(ca, ma)
  return ph, (lambda s: ph.momentum(np.ascontiguousarray(s))[0])


def test_closed_form_coulomb_incline_on_the_kernel_source():
  import closed_form_cases as cf
  from helpers import incline_terrain
  from gym_solo_amd.model import Solo8Model
  ca, ma = make_abi('float64', linear_damping=0.0, angular_damping=0.0)
  n = 6
  e = EmuEngine(ca, ma, n, terrain=incline_terrain(10.0))
  _, mom = _kin(ca, ma)
  mus = cf.incline_frictions(n, seed=5)
  e.params[:, 0] = mus
  e.state[:] = cf.standing_on_incline(n)
  zero = np.zeros((n, 12))
  slides = mus < np.tan(cf.THETA)
  e.rollout(np.zeros((250, n, 12)), abi.STEP_PHYSICS)
  for k in range(3):
    pre = e.state.copy()
    e.step(zero, abi.STEP_PHYSICS)
    for i in range(n):
      got, want = cf.check_coulomb_step(mom, pre[i], e.state[i], mus[i], ca.dt)
      if slides[i]:
        assert mom(e.state[i]) @ cf.T1_SLOPE < -0.05
        assert abs(got - want) < 1e-11 * abs(want), (i, got, want)
      else:
        assert abs(got) < 1e-7, (i, got)


def test_closed_form_motor_clamp_and_push_out_on_the_kernel_source():
  import closed_form_cases as cf
  ca, ma = make_abi('float64', gravity=(0., 0., 0.), linear_damping=0.0, angular_damping=0.0)
  n = 24
  ph, _ = _kin(ca, ma)
  st, acts, far, sign = cf.floating_at_rest(n)
  e = EmuEngine(ca, ma, n)
  e.state[:] = st
  e.step(acts, abi.STEP_PHYSICS)
  for i in range(n):
    M = np.array(ph.step_debug(st[i].copy(), np.zeros(8)).M).reshape(abi.NV, abi.NV)
    base, sat, hold, _, _ = cf.check_motor_clamp(M, st[i], e.state[i], far[i], sign[i], ca.motor_torque_limit * ca.dt)
    assert max(base, sat, hold) < 1e-14, (i, base, sat, hold)
  ca, ma = make_abi('float64')
  st, acts, d, centres, radius = cf.belly_corner_penetrating(n)
  e = EmuEngine(ca, ma, n)
  e.state[:] = st
  e.step(acts, abi.STEP_PHYSICS)
  v = np.array([cf.contact_point_velocity(e.state[i], st[i], centres[i], radius) for i in range(n)])
  np.testing.assert_allclose(v[:, 2], ca.contact_erp * d / ca.dt, rtol=0, atol=1e-11)
  assert np.abs(v[:, :2]).max() < 1e-11


@pytest.mark.parametrize('leg_mu,base_mu,slides', [(0.1, 0.5, False), (0.9, 0.1, True)])
def test_base_link_friction_on_the_kernel_source(leg_mu, base_mu, slides):
  """the base link's spheres keep SoloConfig::base_lateral_friction (solo8v2vanilla.py:157-163 never reaches link -1): per-lane
  friction coefficient in the slot-space solver (f64) and in lane = row (f32) - both against the oracle, f64 against Coulomb"""
  import closed_form_cases as cf
  from helpers import incline_terrain
  from oracle import solo_oracle as so
  for dtype, tol in (('float64', 1e-9), ('float32', 2e-3)):
    ca, ma = make_abi(dtype, lateral_friction=leg_mu, base_lateral_friction=base_mu, linear_damping=0.0, angular_damping=0.0)
    terr = incline_terrain(10.0)
    e = EmuEngine(ca, ma, 2, terrain=terr)
    ph = so.OraclePhysics(ca, ma, terrain=terr)
    mom = lambda s: ph.momentum(np.ascontiguousarray(s))[0]
    st, acts = cf.belly_on_incline(2)
    e.state[:] = st
    ref = st.copy()
    for k in range(120):
      pre = e.state.copy()
      e.step(acts, abi.STEP_PHYSICS)
      ph.step(ref, acts)
    np.testing.assert_allclose(e.state[:, :29], ref[:, :29], rtol=0, atol=tol)
    if dtype == 'float64':
      got, want = cf.check_coulomb_step(mom, pre[0], e.state[0], base_mu, ca.dt)
      assert (abs(got - want) < 1e-11 * abs(want)) if slides else (abs(got) < 1e-10)
      assert (mom(e.state[0]) @ cf.T1_SLOPE < -0.05) == slides


def test_closed_form_joint_limit_and_damping_on_the_kernel_source():
  import closed_form_cases as cf
  from gym_solo_amd.model import Solo8Model
  ca, ma = make_abi('float64', gravity=(0., 0., 0.), linear_damping=0.0, angular_damping=0.0, motor_torque_limit=0.0)
  far = Solo8Model().to_abi()
  for j in range(abi.NUM_DOF):
    far.joint_lower[j], far.joint_upper[j] = -1e3, 1e3
  n = 16
  ph, _ = _kin(ca, ma)
  st, dof, side, c, s = cf.joints_running_into_limits(n, margin=ca.joint_limit_margin)
  e, f = EmuEngine(ca, ma, n), EmuEngine(ca, far, n)
  e.state[:] = st; f.state[:] = st
  zero = np.zeros((n, 12))
  e.step(zero, abi.STEP_PHYSICS); f.step(zero, abi.STEP_PHYSICS)
  for i in range(n):
    M = np.array(ph.step_debug(st[i].copy(), np.zeros(8)).M).reshape(abi.NV, abi.NV)
    rate, off, sign, on = cf.check_joint_limit_against_free(M, st[i], e.state[i], f.state[i], dof[i], side[i], c[i], ca.dt)
    assert rate < 1e-10 and off < 1e-13 and sign == 0.0, (i, rate, off, sign)
  ca, ma = make_abi('float64', gravity=(0., 0., 0.))
  st, acts = cf.translating_afloat(n)
  e = EmuEngine(ca, ma, n)
  e.state[:] = st
  e.step(acts, abi.STEP_PHYSICS)
  v0 = st[:, abi.S_LINVEL:abi.S_LINVEL + 3]
  want = v0 * (1 - ca.dt * ca.linear_damping * (1 + np.linalg.norm(v0, axis=1, keepdims=True)))
  np.testing.assert_allclose(e.state[:, abi.S_LINVEL:abi.S_LINVEL + 3], want, rtol=0, atol=1e-13)
  assert np.abs(e.state[:, abi.S_ANGVEL:abi.S_ANGVEL + 3]).max() < 1e-13 and np.abs(e.state[:, abi.S_QD:abi.S_QD + 8]).max() < 1e-13
