"""The reference-GENERATED golden vectors (tests/golden/obs_reward_golden.*: outputs of the
reference's own obs.py / rewards.py / termination.py, see tests/golden/make_golden.py) pushed
straight through the product kernels: the golden states are written into the engine's state
buffer, the host factories compile the observation / reward / termination programs, and one
``engine.step(None, STEP_OBS | STEP_REWARD)`` launch (no physics) evaluates them.

Shared by the CPU suite (product kernel source on the wave emulator) and the GPU suite (HIP
engine through the C-ABI); each test body takes ``make_env(config=..., **kw)``.

Reference lines exercised: TorsoIMU degrees / clip variants obs.py:277-282, MotorEncoder
max_rotation / degrees obs.py:356-362, normalisation with float32 Box bounds obs.py:149-152,
UprightReward rewards.py:221-234, the margin == 0 step branch of gaussian rewards.py:420-421,
the weighted factory sum rewards.py:104-118, Additive / Multiplicitive rewards.py:146-186,
short-circuit OR over stateful terminations termination.py:46-48.
"""
import json
import os

import numpy as np

from gym_solo_amd import abi
from gym_solo_amd.core import obs as solo_obs
from gym_solo_amd.core import rewards
from gym_solo_amd.core import termination as terms
from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def gold():
  return np.load(os.path.join(G, 'obs_reward_golden.npz'))


def gold_js():
  return json.load(open(os.path.join(G, 'obs_reward_golden.json')))


def golden_state(g):
  """[256, 32] env records (include/solo_engine.h SOLO_S_*) of the fixture's inputs."""
  n = g['quat'].shape[0]
  st = np.zeros((n, abi.STATE_STRIDE))
  st[:, abi.S_POS:abi.S_POS + 3] = g['pos']
  st[:, abi.S_QUAT:abi.S_QUAT + 4] = g['quat']
  st[:, abi.S_ANGVEL:abi.S_ANGVEL + 3] = g['v_ang']
  st[:, abi.S_LINVEL:abi.S_LINVEL + 3] = g['v_lin']
  for d in range(abi.NUM_DOF):
    j = 3 * (d // 2) + d % 2
    st[:, abi.S_Q + d] = g['q'][:, j]
    st[:, abi.S_QD + d] = g['qd'][:, j]
  return st


# the same constructor arguments tests/golden/make_golden.py passed to the reference's classes
OBS = {
  'imu_rad': lambda r: [solo_obs.TorsoIMU(r)],
  'imu_deg': lambda r: [solo_obs.TorsoIMU(r, degrees=True, max_lin_velocity=5, max_angular_velocity=200.)],
  'enc_rad': lambda r: [solo_obs.MotorEncoder(r)],
  'enc_deg_clip': lambda r: [solo_obs.MotorEncoder(r, degrees=True, max_rotation=100.)],
  'enc_clip': lambda r: [solo_obs.MotorEncoder(r, max_rotation=3.0)],
  'bench': lambda r: [solo_obs.TorsoIMU(r), solo_obs.MotorEncoder(r)],
}


def _composite(env):
  # examples/solo8_vanilla/interactive_pos_control.py:22-35
  r = env.robot
  stand = rewards.AdditiveReward()
  stand.client = env.client
  stand.add_term(0.5, rewards.FlatTorsoReward(r, hard_margin=.1, soft_margin=np.pi))
  stand.add_term(0.5, rewards.TorsoHeightReward(r, 0.33698, 0.025, 0.15))
  return [(1, rewards.MultiplicitiveReward(1, stand, rewards.SmallControlReward(r, margin=10),
                                           rewards.HorizontalMoveSpeedReward(r, 0, hard_margin=.5, soft_margin=3)))]


REW = {
  'upright': lambda e: [(1, rewards.UprightReward(e.robot))],
  'flat_torso': lambda e: [(1, rewards.FlatTorsoReward(e.robot, hard_margin=.1, soft_margin=np.pi))],
  'flat_torso_default': lambda e: [(1, rewards.FlatTorsoReward(e.robot))],
  'torso_height': lambda e: [(1, rewards.TorsoHeightReward(e.robot, 0.33698, 0.025, 0.15))],
  'small_control': lambda e: [(1, rewards.SmallControlReward(e.robot, margin=10))],
  'small_control_default': lambda e: [(1, rewards.SmallControlReward(e.robot))],
  'horizontal_speed': lambda e: [(1, rewards.HorizontalMoveSpeedReward(e.robot, 0, hard_margin=.5, soft_margin=3))],
  'horizontal_speed_1': lambda e: [(1, rewards.HorizontalMoveSpeedReward(e.robot, 1, hard_margin=.1, soft_margin=.5))],
  'hard_step': lambda e: [(1, rewards.TorsoHeightReward(e.robot, 0.3, 0.1, 0.0))],   # margin == 0
  'composite': _composite,
  'weighted3': lambda e: [(0.25, rewards.UprightReward(e.robot)), (-2.0, rewards.SmallControlReward(e.robot, margin=10)),
                          (3.0, rewards.TorsoHeightReward(e.robot, 0.33698, 0.025, 0.15))],
}


def _np(t):
  return t.detach().cpu().numpy() if hasattr(t, 'detach') else np.asarray(t)


def _env(make_env, dtype, n, **kw):
  cfg = Solo8VanillaConfig()
  cfg.dtype, cfg._dtype_pinned, cfg.num_envs, cfg._num_envs_pinned = dtype, True, n, True
  cfg.settle_steps = 0  # the state is overwritten with the fixture's inputs
  return make_env(config=cfg, **kw)


def _load_state(env, st):
  import torch
  s = env.engine.state
  s.copy_(torch.as_tensor(st).to(device=s.device, dtype=s.dtype))
  env.client.state_version += 1


def _gimbal_weight(g):
  """How ill-conditioned the Euler angles of each golden orientation are: 1 / sqrt(1 - sarg^2)
  (d asin / d sarg), for the f32 tolerance of angle-derived outputs."""
  q = g['quat']
  sarg = -2 * (q[:, 0] * q[:, 2] - q[:, 3] * q[:, 1])
  return 1.0 / np.sqrt(np.maximum(1e-10, 1.0 - np.minimum(1.0, sarg * sarg)))


def tolerances(dtype, g, expected, angle_derived):
  """f64: 1e-12 absolute on O(1..200) values (the kernel normalises with one fma, the euler angles
  come from the device libm).  f32: 4 ulp-ish of the value's magnitude, and for outputs derived from
  the Euler angles the conditioning of asin / atan2 near the gimbal poles on top."""
  if dtype == 'float64':
    return np.full(expected.shape, 1e-12) * np.maximum(1.0, np.abs(expected))
  tol = 2e-6 * np.maximum(1.0, np.abs(expected))
  if angle_derived:
    w = _gimbal_weight(g)
    w = w.reshape((-1,) + (1,) * (expected.ndim - 1))
    tol = tol + 1e-6 * w * np.maximum(1.0, np.abs(expected))
  return tol


def case_observations(make_env, name, dtype, normalize):
  g = gold()
  st = golden_state(g)
  env = _env(make_env, dtype, st.shape[0], normalize_observations=normalize)
  for o in OBS[name](env.robot):
    env.obs_factory.register_observation(o)
  env._ensure_program()
  assert env._fused['obs']
  _load_state(env, st)
  env.engine.step(None, abi.STEP_OBS)  # the product kernel, observations only
  got = _np(env.engine.obs).astype(np.float64)
  want = g[('obsn_' if normalize else 'obs_') + name]
  assert got.shape == want.shape
  tol = tolerances(dtype, g, want, angle_derived=name.startswith('imu') or name == 'bench')
  err = np.abs(got - want)
  assert (err <= tol).all(), 'max err %.3g (tol %.3g) at %s' % (
    err.max(), tol.flat[err.argmax()], np.unravel_index(err.argmax(), err.shape))
  # the pull-style factory call (obs.py:130-159) goes through the same launch
  vals, labels = env.obs_factory.get_obs()
  np.testing.assert_array_equal(_np(vals).astype(np.float64), got)
  assert len(labels) == want.shape[1]
  env._close()


def case_reward(make_env, name, dtype):
  g = gold()
  st = golden_state(g)
  env = _env(make_env, dtype, st.shape[0])
  for w, r in REW[name](env):
    env.reward_factory.register_reward(w, r)
  env._ensure_program()
  assert env._fused['reward']
  _load_state(env, st)
  env.engine.step(None, abi.STEP_REWARD)
  got = _np(env.engine.reward).astype(np.float64)
  want = g['rew_' + name]
  angle = name in ('upright', 'flat_torso', 'flat_torso_default', 'composite', 'weighted3')
  tol = tolerances(dtype, g, want, angle_derived=angle)
  if dtype == 'float32' and name == 'hard_step':
    # a step function: inputs within f32 rounding of an edge may legitimately land on the other side
    z = g['pos'][:, 2]
    edge = np.minimum(np.abs(z - 0.2), np.abs(z - 0.4)) < 1e-6
    tol = np.where(edge, 1.0, tol)
  err = np.abs(got - want)
  assert (err <= tol).all(), 'max err %.3g at env %d (want %r got %r)' % (
    err.max(), err.argmax(), want[err.argmax()], got[err.argmax()])
  env._close()


def case_terminations(make_env, dtype='float64'):
  """termination.py:38-83 through STEP_DONE-only launches, counters read back from the device:
  every is_terminated() call ticks (as in the reference), later terminations are not ticked once an
  earlier one fires, and a query never mutates the physics state - with auto_reset on too."""
  seqs = gold_js()['termination']
  for auto_reset in (False, True):
    for max_delta in (0, 1, 3):
      cfg = Solo8VanillaConfig()
      cfg.dtype, cfg._dtype_pinned, cfg.num_envs, cfg._num_envs_pinned, cfg.settle_steps = dtype, True, 3, True, 0
      cfg.auto_reset = auto_reset
      env = make_env(config=cfg)
      env.termination_factory.register_termination(terms.TimeBasedTermination(max_delta))
      state0 = _np(env.engine.state).copy()
      state0[:, 0] += 0.25  # off the snapshot, so that an (illegitimate) auto-reset would be seen
      _load_state(env, state0)
      got = []
      for _ in range(6):
        d = _np(env.termination_factory.is_terminated())
        assert d.shape == (3,) and (d == d[0]).all()
        got.append(bool(d[0]))
      assert got == seqs['time_%d' % max_delta]
      np.testing.assert_array_equal(_np(env.engine.state), state0)
      env._close()
    cfg = Solo8VanillaConfig()
    cfg.dtype, cfg._dtype_pinned, cfg.num_envs, cfg._num_envs_pinned, cfg.settle_steps = dtype, True, 2, True, 0
    cfg.auto_reset = auto_reset
    env = make_env(config=cfg)
    env.termination_factory.register_termination(terms.TimeBasedTermination(2), terms.TimeBasedTermination(4))
    trace = []
    for _ in range(8):
      d = _np(env.termination_factory.is_terminated())
      c = _np(env.engine.term_count)
      assert (c == c[0]).all()
      trace.append([bool(d[0]), int(c[0, 0]), int(c[0, 1])])
    assert trace == seqs['factory_2_4']
    env._close()
    env = make_env(config=cfg)
    env.termination_factory.register_termination(terms.PerpetualTermination(), terms.TimeBasedTermination(2))
    got = []
    for _ in range(5):
      d = _np(env.termination_factory.is_terminated())
      got.append([bool(d[0]), int(_np(env.engine.term_count)[0, 1])])
    assert got == seqs['factory_perpetual_2']
    env._close()
