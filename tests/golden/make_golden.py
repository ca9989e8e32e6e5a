"""Generates the committed golden fixtures from the REFERENCE itself (run in the build
container only; /root/reference never travels to the GPU box).

  python tests/golden/make_golden.py [/root/reference]

1. ``joint_info_fixture.json`` — the data literals the reference's own tests hold:
   the 12-row ``getJointInfo`` dump (gym_solo/core/test_obs_observations.py:123-162) and the
   "real case extracted from pybullet" ``getJointState`` rows (:256-275), extracted from the
   test file's AST (data only, no code).
2. ``obs_reward_golden.npz`` / ``obs_reward_golden.json`` — outputs of the reference's own
   pure-Python reductions (gym_solo/core/obs.py, rewards.py, termination.py), imported by file
   path with ``sys.modules`` stubs for the absent third-party modules only (SURVEY.md Appendix
   B) and driven with mock clients exactly as the reference's tests do
   (test_rewards.py:218-224).  ``getEulerFromQuaternion`` is a pybullet C function that is not
   available here: the mock returns the euler angles stored in the fixture's *inputs*
   (computed by oracle/solo_oracle.py:get_euler_from_quaternion, itself pinned by the known
   answer test_obs_observations.py:67-88), so the fixture pins everything downstream of it.
"""
import ast
import importlib.util
import json
import os
import sys
import types
from unittest import mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def extract_literals(ref):
  path = os.path.join(ref, 'gym_solo/core/test_obs_observations.py')
  tree = ast.parse(open(path).read())
  out = {}
  for node in ast.walk(tree):
    if isinstance(node, ast.Assign) and len(node.targets) == 1:
      t = node.targets[0]
      if isinstance(t, ast.Attribute) and t.attr == 'joint_info':
        out['joint_info'] = ast.literal_eval(node.value)
      if isinstance(t, ast.Attribute) and t.attr == 'side_effect' and isinstance(node.value, ast.List) \
         and len(node.value.elts) == 12:
        out['joint_state'] = ast.literal_eval(node.value)
  assert len(out['joint_info']) == 12 and len(out['joint_state']) == 12

  def clean(v):
    if isinstance(v, bytes):
      return v.decode()
    if isinstance(v, (tuple, list)):
      return [clean(x) for x in v]
    return v
  return {'source': {'joint_info': 'gym_solo/core/test_obs_observations.py:123-162',
                     'joint_state': 'gym_solo/core/test_obs_observations.py:256-275'},
          'joint_info': clean(out['joint_info']), 'joint_state': clean(out['joint_state'])}


def load_reference(ref):
  class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
      if shape is not None:
        low = np.full(shape, low)
        high = np.full(shape, high)
      self.low = np.asarray(low, dtype=np.float32)
      self.high = np.asarray(high, dtype=np.float32)
      self.shape = self.low.shape
  gym = types.ModuleType('gym')
  gym.Env = object
  gym.Space = object
  spaces = types.ModuleType('gym.spaces')
  spaces.Box = Box
  spaces.Space = object
  gym.spaces = spaces
  pb = types.ModuleType('pybullet')
  pbu = types.ModuleType('pybullet_utils')
  bc = types.ModuleType('pybullet_utils.bullet_client')
  bc.BulletClient = object
  pbu.bullet_client = bc
  dep = types.ModuleType('deprecation')
  dep.deprecated = lambda **k: (lambda f: f)
  stubs = {'gym': gym, 'gym.spaces': spaces, 'pybullet': pb, 'pybullet_utils': pbu,
           'pybullet_utils.bullet_client': bc, 'deprecation': dep,
           'gym_solo': types.ModuleType('gym_solo'),
           'gym_solo.core': types.ModuleType('gym_solo.core')}
  sys.modules.update(stubs)
  mods = {}
  for name, rel in (('gym_solo.solo_types', 'gym_solo/solo_types.py'),
                    ('gym_solo.core.termination', 'gym_solo/core/termination.py'),
                    ('gym_solo.core.rewards', 'gym_solo/core/rewards.py'),
                    ('gym_solo.core.obs', 'gym_solo/core/obs.py')):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ref, rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    setattr(sys.modules['gym_solo'], name.split('.')[-1], m)
    mods[name.split('.')[-1]] = m
  return mods


def main():
  sys.dont_write_bytecode = True
  ref = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
  with open(os.path.join(HERE, 'joint_info_fixture.json'), 'w') as f:
    json.dump(extract_literals(ref), f, indent=1)

  from oracle import solo_oracle as so  # only for the euler inputs (see module docstring)
  mods = load_reference(ref)
  obs, rewards, termination = mods['obs'], mods['rewards'], mods['termination']

  rng = np.random.default_rng(0)
  n = 256
  quat = rng.normal(size=(n, 4))
  quat /= np.linalg.norm(quat, axis=1, keepdims=True)
  quat[:8] = [[0, 0, 0, 1], [0, 0, .707, .707], [0, .70710678, 0, .70710678],
              [0, -.70710678, 0, .70710678], [.5, .5, .5, .5], [0.05, 0.1, 0, 0.99],
              [0, 0.7071, 0, 0.7071], [1, 0, 0, 0]]
  euler = so.get_euler_from_quaternion(quat)
  pos = rng.uniform(-1, 1, (n, 3))
  pos[:, 2] = rng.uniform(0, 0.6, n)
  v_lin = rng.normal(scale=4, size=(n, 3))
  v_lin[:16] *= 0.05
  v_ang = rng.normal(scale=6, size=(n, 3))
  q = rng.uniform(-2 * np.pi, 2 * np.pi, (n, 12))
  qd = rng.normal(scale=8, size=(n, 12))
  qd[:16] *= 0.05
  q[:, 2::3] = 0
  qd[:, 2::3] = 0
  inputs = dict(quat=quat, euler=euler, pos=pos, v_lin=v_lin, v_ang=v_ang, q=q, qd=qd)

  def client_for(i):
    c = mock.MagicMock()
    c.getBasePositionAndOrientation.return_value = (tuple(pos[i]), tuple(quat[i]))
    c.getEulerFromQuaternion.return_value = tuple(euler[i])
    c.getBaseVelocity.return_value = (tuple(v_lin[i]), tuple(v_ang[i]))
    c.getNumJoints.return_value = 12
    c.getJointState.side_effect = lambda robot, j: (q[i, j], qd[i, j], (0,) * 6, 0.0)
    c.getJointInfo.side_effect = lambda robot, j: (j, b'', 0, 0, 0, 0, 0., 0., -10.0, 10.0)
    return c

  out = dict(inputs)
  obs_cases = {
    'imu_rad': lambda: obs.TorsoIMU(0),
    'imu_deg': lambda: obs.TorsoIMU(0, degrees=True, max_lin_velocity=5, max_angular_velocity=200.),
    'enc_rad': lambda: obs.MotorEncoder(0),
    'enc_deg_clip': lambda: obs.MotorEncoder(0, degrees=True, max_rotation=100.),
    'enc_clip': lambda: obs.MotorEncoder(0, max_rotation=3.0),
  }
  for name, make in obs_cases.items():
    raw, norm = [], []
    for i in range(n):
      c = client_for(i)
      for normalize, dst in ((False, raw), (True, norm)):
        f = obs.ObservationFactory(c, normalize=normalize)
        f.register_observation(make())
        dst.append(f.get_obs()[0])
    out['obs_' + name] = np.array(raw)
    out['obsn_' + name] = np.array(norm)
  # the benchmark observation: TorsoIMU + MotorEncoder, concatenated (test_solo8v2vanilla.py:179-180)
  both, bothn = [], []
  for i in range(n):
    c = client_for(i)
    for normalize, dst in ((False, both), (True, bothn)):
      f = obs.ObservationFactory(c, normalize=normalize)
      f.register_observation(obs.TorsoIMU(0))
      f.register_observation(obs.MotorEncoder(0))
      dst.append(f.get_obs()[0])
  out['obs_bench'] = np.array(both)
  out['obsn_bench'] = np.array(bothn)

  def composite(c):
    # examples/solo8_vanilla/interactive_pos_control.py:22-35
    flat = rewards.FlatTorsoReward(0, hard_margin=.1, soft_margin=np.pi)
    height = rewards.TorsoHeightReward(0, 0.33698, 0.025, 0.15)
    small = rewards.SmallControlReward(0, margin=10)
    no_move = rewards.HorizontalMoveSpeedReward(0, 0, hard_margin=.5, soft_margin=3)
    stand = rewards.AdditiveReward()
    stand.client = c
    stand.add_term(0.5, flat)
    stand.add_term(0.5, height)
    home = rewards.MultiplicitiveReward(1, stand, small, no_move)
    f = rewards.RewardFactory(c)
    f.register_reward(1, home)
    return f

  reward_cases = {
    'upright': lambda c: rewards.UprightReward(0),
    'flat_torso': lambda c: rewards.FlatTorsoReward(0, hard_margin=.1, soft_margin=np.pi),
    'flat_torso_default': lambda c: rewards.FlatTorsoReward(0),
    'torso_height': lambda c: rewards.TorsoHeightReward(0, 0.33698, 0.025, 0.15),
    'small_control': lambda c: rewards.SmallControlReward(0, margin=10),
    'small_control_default': lambda c: rewards.SmallControlReward(0),
    'horizontal_speed': lambda c: rewards.HorizontalMoveSpeedReward(0, 0, hard_margin=.5, soft_margin=3),
    'horizontal_speed_1': lambda c: rewards.HorizontalMoveSpeedReward(0, 1, hard_margin=.1, soft_margin=.5),
    'hard_step': lambda c: rewards.TorsoHeightReward(0, 0.3, 0.1, 0.0),
  }
  for name, make in reward_cases.items():
    vals = []
    for i in range(n):
      c = client_for(i)
      r = make(c)
      r.client = c
      vals.append(r.compute())
    out['rew_' + name] = np.array(vals, dtype=np.float64)
  out['rew_composite'] = np.array([composite(client_for(i)).get_reward() for i in range(n)])
  # weighted factory of three terms (rewards.py:104-118)
  vals = []
  for i in range(n):
    c = client_for(i)
    f = rewards.RewardFactory(c)
    f.register_reward(0.25, rewards.UprightReward(0))
    f.register_reward(-2.0, rewards.SmallControlReward(0, margin=10))
    f.register_reward(3.0, rewards.TorsoHeightReward(0, 0.33698, 0.025, 0.15))
    vals.append(f.get_reward())
  out['rew_weighted3'] = np.array(vals)

  np.savez_compressed(os.path.join(HERE, 'obs_reward_golden.npz'), **out)

  # scalar known answers + termination sequences
  xs = [-3.0, -1.0, -0.25, 0.0, 0.25, 0.5, 1.0, 2.0, 3.0]
  g_cases = [((0., 0.), 1., .25), ((-1., 1.), 1., .25), ((0., 0.), .5, .1), ((0.3, 0.4), 0., .1),
             ((-.1, .1), np.pi, .1)]
  js = {'gaussian': [{'bounds': list(b), 'margin': m, 'margin_value': mv,
                      'x': xs, 'y': [rewards.gaussian(x, b, m, mv) for x in xs]}
                     for b, m, mv in g_cases],
        'gaussian_vector': {'x': [0, .25, 1, 3],
                            'y': list(map(float, rewards.gaussian(np.array([0, .25, 1, 3.]),
                                                                  (0., 0.), 1., .25)))},
        'linear': [{'args': [x, t, s, sym], 'y': float(rewards.linear(x, t, s, sym))}
                   for x in (4., 5., 6., 8., 9., 10.) for t, s in ((5., 4.), (5., 0.), (5., -4.))
                   for sym in (False, True)]}
  seqs = {}
  for max_delta in (0, 1, 3):
    t = termination.TimeBasedTermination(max_delta)
    seqs['time_%d' % max_delta] = [bool(t.is_terminated()) for _ in range(6)]
  # short-circuit OR: second TimeBased is not ticked once the first fires (termination.py:46-48)
  f = termination.TerminationFactory()
  t1, t2 = termination.TimeBasedTermination(2), termination.TimeBasedTermination(4)
  f.register_termination(t1, t2)
  trace = []
  for _ in range(8):
    d = f.is_terminated()
    trace.append([bool(d), t1.step_delta, t2.step_delta])
  seqs['factory_2_4'] = trace
  f = termination.TerminationFactory()
  p, t3 = termination.PerpetualTermination(), termination.TimeBasedTermination(2)
  f.register_termination(p, t3)
  seqs['factory_perpetual_2'] = [[bool(f.is_terminated()), t3.step_delta] for _ in range(5)]
  js['termination'] = seqs
  with open(os.path.join(HERE, 'obs_reward_golden.json'), 'w') as fjs:
    json.dump(js, fjs, indent=1)
  print('wrote fixtures to', HERE)


if __name__ == '__main__':
  main()
