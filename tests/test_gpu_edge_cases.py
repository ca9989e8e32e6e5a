"""Edge cases of the C-ABI on the GPU: single robot, ragged batch / rollout lengths against the
fused-launch and stream-slice sizes, empty rollout, masked reset, a robot driven non-finite, bad
arguments (the reference's own tests cover the 1-robot case only; these are the batched shapes)."""
import numpy as np
import pytest

from gym_solo_amd import abi
from helpers import make_abi, random_actions

pytestmark = pytest.mark.gpu


@pytest.fixture
def torch():
  import torch
  if not torch.cuda.is_available():
    pytest.fail('GPU tests need a visible MI355X')
  return torch


def _engine(n, dtype='float64', **kw):
  from gym_solo_amd.engine import Engine
  ca, ma = make_abi(dtype, **kw)
  return Engine(ca, ma, n), ca, ma


def test_single_robot_matches_oracle(torch):
  from oracle import solo_oracle as so
  eng, ca, ma = _engine(1)
  ph = so.OraclePhysics(ca, ma)
  st = eng.state.cpu().numpy().copy()
  rng = np.random.default_rng(0)
  for k in range(30):
    a = random_actions(rng, 1)
    ph.step(st, a)
    eng.step(torch.as_tensor(a, device='cuda'), abi.STEP_PHYSICS)
  np.testing.assert_allclose(eng.state.cpu().numpy()[:, :29], st[:, :29], rtol=0, atol=1e-9)
  eng.close()


@pytest.mark.parametrize('n,spl,streams,k', [(777, 7, 2, 23), (5, 4, 3, 9), (1, 3, 2, 4), (130, 1, 4, 3)])
def test_ragged_sizes_fused_equals_single_steps(torch, n, spl, streams, k):
  """Batch not divisible by the stream slices, rollout length not divisible by steps_per_launch,
  more slices than make sense for the batch: same trajectories as K single steps, bit for bit."""
  a, _, _ = _engine(n, 'float32', steps_per_launch=spl, rollout_streams=streams, settle_steps=50)
  b, _, _ = _engine(n, 'float32', steps_per_launch=1, rollout_streams=1, settle_steps=50)
  rng = np.random.default_rng(n)
  acts = torch.as_tensor(rng.uniform(-6, 6, (k, n, 12)), device='cuda', dtype=torch.float32)
  a.rollout(acts, abi.STEP_PHYSICS)
  for i in range(k):
    b.step(acts[i], abi.STEP_PHYSICS)
  torch.cuda.synchronize()
  assert torch.equal(a.state, b.state)
  assert torch.equal(a.targets, b.targets)
  a.close(); b.close()


def test_restitution_is_accepted_and_without_effect(torch):
  """configs.py:23 / solo8v2vanilla.py:158-163 pass `restitution` to changeDynamics for the robot's links.  Bullet gives
  a contact the product of its two bodies' restitutions ([recalled] btManifoldResult::calculateCombinedRestitution) and
  the reference's ground (plane.urdf) has none: any value in [0, 1] is accepted and leaves every trajectory as it is -
  robots dropped from the reset height, bit for bit, and against the oracle."""
  from oracle import solo_oracle as so
  a, ca, ma = _engine(64, 'float64', settle_steps=0, restitution=0.8)
  b, _, _ = _engine(64, 'float64', settle_steps=0)
  rng = np.random.default_rng(3)
  acts = torch.as_tensor(rng.uniform(-6, 6, (400, 64, 12)), device='cuda', dtype=torch.float64)
  a.rollout(acts, abi.STEP_PHYSICS)
  b.rollout(acts, abi.STEP_PHYSICS)
  torch.cuda.synchronize()
  assert float(b.state[:, 2].max()) < 0.45     # (they have landed)
  assert torch.equal(a.state, b.state)
  ph = so.OraclePhysics(ca, ma)                # (the oracle is handed the same configuration, restitution included)
  e, _, _ = _engine(2, 'float64', settle_steps=0, restitution=0.8)
  st = e.state.cpu().numpy().copy()
  for k in range(300):
    ph.step(st, acts[k, :2].cpu().numpy())
  e.rollout(acts[:300, :2].contiguous(), abi.STEP_PHYSICS)
  np.testing.assert_allclose(e.state.cpu().numpy()[:, :29], st[:, :29], rtol=0, atol=1e-9)
  with pytest.raises(Exception, match='restitution'):
    _engine(4, 'float64', restitution=1.5)
  a.close(); b.close(); e.close()


def test_empty_rollout_is_a_noop(torch):
  eng, _, _ = _engine(8, 'float32', settle_steps=20)
  before = eng.state.clone()
  eng.rollout(torch.empty(0, 8, 12, device='cuda', dtype=torch.float32), abi.STEP_PHYSICS)
  torch.cuda.synchronize()
  assert torch.equal(eng.state, before)
  eng.close()


def test_masked_reset_restores_only_the_masked_robots(torch):
  eng, _, _ = _engine(16, 'float32', settle_steps=50)
  rng = np.random.default_rng(2)
  for k in range(10):
    eng.step(torch.as_tensor(random_actions(rng, 16), device='cuda', dtype=torch.float32), abi.STEP_PHYSICS)
  moved = eng.state.clone()
  mask = torch.zeros(16, dtype=torch.uint8, device='cuda')
  mask[[1, 5, 15]] = 1
  eng.reset(mask)
  torch.cuda.synchronize()
  sel = mask.bool()
  assert torch.equal(eng.state[sel], eng.snapshot[sel])
  assert torch.equal(eng.state[~sel], moved[~sel])
  eng.reset()
  torch.cuda.synchronize()
  assert torch.equal(eng.state, eng.snapshot)
  eng.close()


def test_non_finite_robot_is_restored_and_counted(torch):
  """A NaN target makes one robot's state non-finite: the kernel restores that robot from its
  snapshot, counts it in the statistics, and its neighbours step on untouched."""
  eng, _, _ = _engine(8, 'float32', settle_steps=50)
  ref, _, _ = _engine(8, 'float32', settle_steps=50)
  rng = np.random.default_rng(3)
  a = random_actions(rng, 8).astype(np.float32)
  bad = a.copy()
  bad[2, 4] = np.nan
  eng.step(torch.as_tensor(bad, device='cuda'), abi.STEP_PHYSICS)
  ref.step(torch.as_tensor(a, device='cuda'), abi.STEP_PHYSICS)
  torch.cuda.synchronize()
  assert torch.isfinite(eng.state).all()
  assert torch.equal(eng.state[2, :29], eng.snapshot[2, :29])
  keep = [i for i in range(8) if i != 2]
  assert torch.equal(eng.state[keep], ref.state[keep])
  assert float(eng.stats[5]) == 1.0
  eng.close(); ref.close()


def test_bad_arguments_raise(torch):
  eng, _, _ = _engine(4, 'float32', settle_steps=10)
  with pytest.raises(ValueError):
    eng.set_params(7, torch.ones(4, device='cuda'))
  with pytest.raises((ValueError, TypeError)):
    eng.step(torch.zeros(3, 12, device='cuda'), abi.STEP_PHYSICS)        # wrong batch size
  # (another FLOATING precision is converted by the binding - the drop-in env computes in float64 by default and callers
  # hand it float32 actions -; a non-floating dtype is an error)
  eng.step(torch.zeros(4, 12, device='cuda', dtype=torch.float64), abi.STEP_PHYSICS)
  with pytest.raises((ValueError, TypeError)):
    eng.step(torch.zeros(4, 12, device='cuda', dtype=torch.int32), abi.STEP_PHYSICS)  # wrong dtype
  with pytest.raises(ValueError):
    eng.step(torch.zeros(4, 12, device='cuda'), abi.STEP_ALL)             # no program registered
  eng.close()


def test_ragged_recording_rollout_with_slices_equals_single_steps(torch):
  """Everything at once: 777 robots on 3 stream slices, 23 steps in fused launches of 7 (the last
  one of 2), auto-reset inside a launch, every step's observation / reward / done recorded by the
  output epilogue - identical to 23 single-step launches (which evaluate their outputs in place)."""
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig, Solo8VanillaEnv
  from gym_solo_amd.workloads import register_benchmark_workload
  out = {}
  for spl, streams in ((1, 1), (7, 3)):
    cfg = Solo8VanillaConfig()
    cfg.dtype, cfg.auto_reset, cfg.steps_per_launch, cfg.rollout_streams = 'float32', True, spl, streams
    cfg.num_envs, cfg.settle_steps = 777, 60
    env = Solo8VanillaEnv(config=cfg)
    register_benchmark_workload(env, max_steps=9)
    env._ensure_program()
    g = torch.Generator(device='cuda').manual_seed(3)
    acts = (torch.rand(23, 777, 12, device='cuda', dtype=torch.float32, generator=g) * 2 - 1) * 6.28
    if spl == 1:
      obs, rew, done = [], [], []
      for k in range(23):
        env.engine.step(acts[k], abi.STEP_ALL)
        obs.append(env.engine.obs.clone()); rew.append(env.engine.reward.clone()); done.append(env.engine.done.clone())
      rec = (torch.stack(obs), torch.stack(rew), torch.stack(done))
    else:
      rec = env.engine.rollout(acts, abi.STEP_ALL, record=True)
    env.engine.synchronize()
    out[spl] = [t.cpu().numpy() for t in rec] + [env.engine.state.cpu().numpy(), env.engine.term_count.cpu().numpy(),
                                                 env.engine.stats.cpu().numpy(), env.engine.obs.cpu().numpy(),
                                                 env.engine.reward.cpu().numpy()]
    env._close()
  for i, (a, b) in enumerate(zip(out[1], out[7])):
    if i == 5:  # episodic statistics: double atomics from 12 robots per shard, in scheduling order
      np.testing.assert_allclose(a, b, rtol=1e-12, atol=0)
    else:
      np.testing.assert_array_equal(a, b)
  assert out[1][2].sum() == 2 * 777  # two episode ends (steps 10 and 20) per robot


@pytest.mark.parametrize('resid,on_limit', [(0.0, 1e-9), (1e-7, 1e-6)])
def test_joint_limit_rows_match_oracle_gpu(torch, resid, on_limit):
  """URDF joint limits (+-10 rad, test_obs_observations.py:123-162 cols 8-9) on the HIP engine:
  joints driven towards a limit with targets beyond it stop ON the limit (to 1e-9 rad when the solver runs to its
  fixed point, to 1e-6 with pybullet's default residual threshold); f64 engine == oracle, and the f32 engine
  respects the limit too."""
  from helpers import joint_limit_case
  from oracle import solo_oracle as so
  eng, ca, ma = _engine(8, solver_residual_threshold=resid)
  ph = so.OraclePhysics(ca, ma)
  st, tg = joint_limit_case(ph, n=8, seed=3)
  eng.state.copy_(torch.as_tensor(st, device='cuda'))
  e32, _, _ = _engine(8, 'float32', solver_residual_threshold=resid)
  e32.state.copy_(torch.as_tensor(st, device='cuda', dtype=torch.float32))
  for k in range(40):
    ph.step(st, tg)
    eng.step(torch.as_tensor(tg, device='cuda'), abi.STEP_PHYSICS)
    e32.step(torch.as_tensor(tg, device='cuda', dtype=torch.float32), abi.STEP_PHYSICS)
    assert np.abs(st[:, abi.S_Q:abi.S_Q + 8]).max() <= 10.0 + on_limit
  q = eng.state.cpu().numpy()[:, abi.S_Q:abi.S_Q + 8]
  assert ((np.abs(q) > 10.0 - on_limit).any(axis=1)).all() and np.abs(q).max() <= 10.0 + on_limit
  np.testing.assert_allclose(eng.state.cpu().numpy()[:, :29], st[:, :29], rtol=0, atol=1e-9)
  assert float(e32.state[:, abi.S_Q:abi.S_Q + 8].abs().max()) <= 10.0 + 1e-5
  eng.close(); e32.close()


def test_engines_are_freed_without_close_and_guard_after_close(torch):
  """Engine lifetime: an Engine that is never close()d is collected (its zero-copy views hold no
  reference back to it) and its device buffers are freed; after close() the C-ABI calls raise
  instead of touching freed memory."""
  import gc
  from gym_solo_amd.engine import Engine, EngineError
  ca, ma = make_abi('float32', steps_per_launch=250, settle_steps=2)
  gc.collect(); torch.cuda.synchronize()
  free0, _ = torch.cuda.mem_get_info()
  for _ in range(12):                       # ~150 MB each (state, traj [250][4096][32], scratch)
    eng = Engine(ca, ma, 4096)
    view = eng.state                        # a zero-copy view handed out
    del view
    del eng
    gc.collect()
  torch.cuda.synchronize()
  free1, _ = torch.cuda.mem_get_info()
  assert free0 - free1 < 200 * 2 ** 20, 'engines leaked: %d MiB' % ((free0 - free1) >> 20)
  eng = Engine(ca, ma, 8)
  eng.close()
  assert eng.is_closed and eng.state is None
  with pytest.raises(EngineError):
    eng.step(None, abi.STEP_PHYSICS)
  eng.close()  # idempotent


@pytest.mark.parametrize('streams', [1, 2])
def test_launch_order_does_not_change_results(torch, streams):
  """solo_engine_set_order / Engine.balance(): a permuted workgroup -> robot mapping (cost-balanced
  scheduling) leaves every robot's trajectory, outputs, counters and statistics bit-identical, with
  and without stream slices; view.cost holds the sweeps of the last launch."""
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig, Solo8VanillaEnv
  from gym_solo_amd.workloads import register_benchmark_workload
  out = {}
  n = 96
  for mode in ('identity', 'random', 'balanced'):
    cfg = Solo8VanillaConfig()
    cfg.num_envs, cfg.dtype, cfg.auto_reset, cfg.steps_per_launch, cfg.rollout_streams = n, 'float32', True, 8, streams
    env = Solo8VanillaEnv(config=cfg)
    register_benchmark_workload(env, max_steps=13)
    env._ensure_program()
    eng = env.engine
    g = torch.Generator(device='cuda').manual_seed(5)
    acts = (torch.rand(40, n, 12, device='cuda', generator=g) * 2 - 1) * 6.28
    eng.rollout(acts[:8], abi.STEP_ALL)
    assert int(eng.cost.min()) >= 8 and int(eng.cost.max()) <= 8 * 50   # sweeps of the 8-step launch
    if mode == 'random':
      parts = []
      for s in range(streams):
        lo, hi = n * s // streams, n * (s + 1) // streams
        parts.append(torch.randperm(hi - lo, device='cuda', generator=g).to(torch.int32) + lo)
      eng.set_order(torch.cat(parts))
    elif mode == 'balanced':
      eng.balance()
    rec = eng.rollout(acts[8:], abi.STEP_ALL, record=True)
    eng.synchronize()
    out[mode] = [t.cpu().numpy() for t in rec] + [eng.state.cpu().numpy(), eng.term_count.cpu().numpy(),
                                                  eng.stats.cpu().numpy(), eng.cost.cpu().numpy()]
    eng.set_order(None)
    env._close()
  for mode in ('random', 'balanced'):
    for a, b in zip(out['identity'], out[mode]):
      np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('streams', [1, 2])
def test_invalid_launch_order_is_rejected(torch, streams):
  """solo_engine_set_order validates its table on upload: out-of-range entries, duplicates and - with
  rollout slices - robots that leave their slice are SOLO_ERR_INVALID_ARG (ValueError), and the engine
  keeps stepping with the previous order."""
  eng, ca, ma = _engine(8, 'float32', rollout_streams=streams)
  ident = torch.arange(8, dtype=torch.int32, device='cuda')
  bad = [ident.clone() for _ in range(3)]
  bad[0][3] = 8            # out of range
  bad[1][3] = -1
  bad[2][5] = bad[2][4]    # duplicate
  if streams == 2:
    swapped = ident.clone()
    swapped[0], swapped[7] = 7, 0   # a permutation, but robots cross the slice boundary
    bad.append(swapped)
  for o in bad:
    with pytest.raises(ValueError):
      eng.set_order(o)
  eng.set_order(ident.flip(0) if streams == 1 else torch.cat([ident[:4].flip(0), ident[4:].flip(0)]))
  eng.step(torch.zeros(8, 12, device='cuda'), abi.STEP_PHYSICS)
  eng.synchronize()
  assert bool(torch.isfinite(eng.state).all())
  eng.close()


def test_rejected_terrain_leaves_the_previous_ground_in_force(torch):
  """solo_engine_set_terrain validates its argument BEFORE it touches anything: a degenerate grid or a non-positive
  cell size is SOLO_ERR_INVALID_ARG (ValueError) and the engine keeps the heightfield - and the settle snapshot - it
  had: it goes on stepping exactly like a twin that never saw the bad call."""
  import helpers
  engines = []
  for _ in range(2):
    eng, ca, ma = _engine(8, 'float64')
    eng.set_terrain(helpers.incline_terrain())
    engines.append(eng)
  good = helpers.incline_terrain()
  for field, value in (('nx', 1), ('ny', 1), ('cell', 0.0), ('cell', -1.0)):
    bad = helpers.incline_terrain()
    setattr(bad, field, value)
    with pytest.raises(ValueError):
      engines[0].set_terrain(bad)
  acts = random_actions(np.random.default_rng(5), 8)
  for eng in engines:
    for _ in range(20):
      eng.step(torch.as_tensor(acts, device='cuda'), abi.STEP_PHYSICS)
    eng.synchronize()
  np.testing.assert_array_equal(engines[0].state.cpu().numpy(), engines[1].state.cpu().numpy())
  np.testing.assert_array_equal(engines[0].snapshot.cpu().numpy(), engines[1].snapshot.cpu().numpy())
  flat, _, _ = _engine(8, 'float64')
  assert not np.array_equal(engines[0].snapshot.cpu().numpy(), flat.snapshot.cpu().numpy())   # (the incline IS in force)
  flat.close()
  del good
  for eng in engines:
    eng.close()


def test_fast_spin_takes_the_library_rotation_path(torch):
  """The f32 rotation update uses even Taylor polynomials in (|w| dt / 2)^2 and falls back to the
  library sincos above 1/16 (|w| > 500 rad/s): a base spinning at 300 / 700 rad/s in the air, one
  step, f32 engine vs oracle - both paths agree with the exact update."""
  from oracle import solo_oracle as so
  eng, ca, ma = _engine(4, 'float32')
  ca64, _ = make_abi('float64')
  ph = so.OraclePhysics(ca64, ma)
  st = np.tile(ph.settle(1), (4, 1))
  st[:, abi.S_POS + 2] = 1.0
  st[0, abi.S_ANGVEL:abi.S_ANGVEL + 3] = [300.0, 0.0, 0.0]
  st[1, abi.S_ANGVEL:abi.S_ANGVEL + 3] = [0.0, 700.0, 0.0]
  st[2, abi.S_ANGVEL:abi.S_ANGVEL + 3] = [400.0, -300.0, 500.0]
  eng.state.copy_(torch.as_tensor(st, device='cuda', dtype=torch.float32))
  a = np.zeros((4, 12))
  ph.step(st, a)
  eng.step(torch.zeros(4, 12, device='cuda'), abi.STEP_PHYSICS)
  got = eng.state.cpu().numpy().astype(np.float64)
  np.testing.assert_allclose(got[:, abi.S_QUAT:abi.S_QUAT + 4], st[:, abi.S_QUAT:abi.S_QUAT + 4], rtol=0, atol=2e-5)
  np.testing.assert_allclose(np.linalg.norm(got[:, abi.S_QUAT:abi.S_QUAT + 4], axis=1), 1.0, atol=1e-6)
  eng.close()
