"""GPU-side tests of SURVEY.md §8f rows N3 (URDF -> model loader) and N4 (VectorEnv adapter +
registration) on the HIP engine, and the distribution-level f32 / f64 comparison on the benchmark
workload."""
import numpy as np
import pytest

import env_cases as cases
from test_gpu_env import make_env

pytestmark = pytest.mark.gpu


def test_env_built_from_a_urdf_file_steps_bit_identically(tmp_path):
  """N3 - loadURDF(config.urdf, flags=URDF_USE_INERTIA_FROM_FILE) (solo8v2vanilla.py:151-155,
  configs.py:36-38): an env whose model comes from a URDF FILE, stepped on the GPU next to the
  built-in-model env: identical bits (the file reproduces the C-ABI model exactly)."""
  import torch
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig
  from gym_solo_amd.urdf import UrdfSolo8Model, to_urdf
  path = tmp_path / 'solo.urdf'
  path.write_text(to_urdf())
  envs = []
  for urdf in (str(path), None):
    cfg = Solo8VanillaConfig()
    cfg.urdf_path = urdf
    env = make_env(config=cfg)
    cases.register_benchmark_workload(env, max_steps=1000)
    envs.append(env)
  a, b = envs
  assert isinstance(a.solo_model, UrdfSolo8Model) and not isinstance(b.solo_model, UrdfSolo8Model)
  np.testing.assert_array_equal(cases.np_(a.engine.snapshot), cases.np_(b.engine.snapshot))
  rng = np.random.default_rng(8)
  for k in range(20):
    act = torch.as_tensor(rng.uniform(-2 * np.pi, 2 * np.pi, (a.num_envs, 12)))
    oa, ra, da, _ = a.step(act)
    ob, rb, db, _ = b.step(act)
    np.testing.assert_array_equal(cases.np_(oa), cases.np_(ob))
    np.testing.assert_array_equal(cases.np_(ra), cases.np_(rb))
  np.testing.assert_array_equal(cases.np_(a.engine.state), cases.np_(b.engine.state))
  # a different file really changes the simulation: a heavier base sinks differently
  heavy = tmp_path / 'heavy.urdf'
  heavy.write_text(to_urdf().replace('<mass value="1.16115091"/>', '<mass value="2.0"/>'))
  cfg = Solo8VanillaConfig()
  cfg.urdf_path = str(heavy)
  c = make_env(config=cfg)
  assert np.abs(cases.np_(c.engine.snapshot)[:, :29] - cases.np_(b.engine.snapshot)[:, :29]).max() > 1e-6


def test_vector_env_adapter_on_the_hip_engine():
  """N4 - gym_solo/__init__.py:3-11 ids through gym_solo_amd.make + the VectorEnv-style adapter
  over the in-kernel auto-reset: 5-tuple semantics, time limit -> truncated, the next observation
  after a truncation comes from the restored state, zero-copy aliasing of the engine's buffers."""
  import torch
  import gym_solo_amd
  from gym_solo_amd.core import obs as solo_obs, termination as terms
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig
  from gym_solo_amd.testing import DummyTermination
  from gym_solo_amd.vector import Solo8VectorEnv
  cfg = Solo8VanillaConfig()
  cfg.dtype, cfg.num_envs, cfg.auto_reset = 'float64', 32, True
  env = gym_solo_amd.make('solo8vanilla-v0', config=cfg, copy_outputs=False)
  env.obs_factory.register_observation(solo_obs.TorsoIMU(env.robot))
  env.obs_factory.register_observation(solo_obs.MotorEncoder(env.robot))
  from gym_solo_amd.core import rewards
  env.reward_factory.register_reward(1, rewards.TorsoHeightReward(env.robot, 0.33698, 0.025, 0.15))
  env.termination_factory.register_termination(terms.TimeBasedTermination(3))
  venv = Solo8VectorEnv(env)
  assert venv.num_envs == 32 and venv.observation_space.shape == (32, 21) and venv.action_space.shape == (32, 12)
  obs0, info = venv.reset(seed=1)
  obs0 = cases.np_(obs0).copy()
  assert obs0.shape == (32, 21) and info == {}
  rng = np.random.default_rng(0)
  first_after_reset = None
  for k in range(9):
    act = torch.as_tensor(rng.uniform(-2, 2, (32, 12)), device='cuda')
    o, r, terminated, truncated, info = venv.step(act)
    assert o.data_ptr() == env.engine.obs.data_ptr() and r.data_ptr() == env.engine.reward.data_ptr()  # zero copy
    assert not bool(terminated.any())
    assert bool(truncated.all()) == (k % 4 == 3) and bool(truncated.any()) == (k % 4 == 3)
    assert info['labels'][:3] == ['θx', 'θy', 'θz']
    if k % 4 == 3:
      # the robots were restored in-kernel: the state IS the snapshot again
      np.testing.assert_array_equal(cases.np_(env.engine.state)[:, :29], cases.np_(env.engine.snapshot)[:, :29])
      np.testing.assert_array_equal(cases.np_(venv.env.obs_factory.get_obs()[0]), obs0)
  # a real termination (not a time limit) is reported as `terminated`
  cfg2 = Solo8VanillaConfig()
  cfg2.dtype, cfg2.num_envs, cfg2.auto_reset = 'float32', 8, True
  env2 = gym_solo_amd.make('solo8vanilla-v0', config=cfg2)
  env2.obs_factory.register_observation(solo_obs.TorsoIMU(env2.robot))
  env2.reward_factory.register_reward(1, rewards.UprightReward(env2.robot))
  env2.termination_factory.register_termination(DummyTermination(0, True))
  o, r, terminated, truncated, _ = Solo8VectorEnv(env2).step(torch.zeros(8, 12))
  assert bool(terminated.all()) and not bool(truncated.any())
  cfg3 = Solo8VanillaConfig()
  cfg3.num_envs = 4
  with pytest.raises(ValueError):
    Solo8VectorEnv(gym_solo_amd.make('solo8vanilla-v0', config=cfg3))


def test_f32_and_f64_engines_simulate_the_same_system_statistically():
  """The benchmark workload is chaotic (f32 and f64 trajectories decorrelate after ~200 steps), so
  the f32 headline is compared with the f64 parity engine at the level of distributions: 4096
  robots x 1000 steps (one full episode each), the same U(-2pi, 2pi) action stream through both
  engines, every step's observation / reward recorded.  Two independent realisations of the same
  chaotic system differ by sampling noise (std of the mean return over 4096 episodes = 0.25 %), so
  the bounds are a few sigma of that: episodic-return mean within 1 % (measured 0.003 ... 0.06 %), its
  standard deviation within 5 % (0.3 ... 0.8 %), per-step mean reward within 1 %, mean |roll, pitch|,
  mean |joint angle| and mean base speed within 1 % (<= 0.2 %), the late-episode roll histogram
  within 0.01 total variation (0.003), nobody diverged in either engine."""
  import torch
  from gym_solo_amd import abi
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig
  n, k = 4096, 1000
  g = torch.Generator(device='cuda').manual_seed(1234)
  acts64 = (torch.rand(k, n, 12, device='cuda', dtype=torch.float64, generator=g) * 2 - 1) * (2 * np.pi)
  out = {}
  for dtype in ('float64', 'float32'):
    cfg = Solo8VanillaConfig()
    cfg.dtype, cfg._dtype_pinned, cfg.num_envs, cfg._num_envs_pinned = dtype, True, n, True
    cfg.auto_reset, cfg.steps_per_launch, cfg.rollout_streams = True, 100, 2
    env = make_env(config=cfg)
    cases.register_benchmark_workload(env, max_steps=k - 1)
    env._ensure_program()
    eng = env.engine
    acts = acts64.to(eng.tdtype)
    obs, rew, done = eng.rollout(acts, abi.STEP_ALL, record=True)
    assert bool(done[-1].all()) and int(done.sum()) == n
    st = eng.stats.cpu().numpy()
    rec = dict(mean_return=st[0] / st[2], std_return=np.sqrt(max(0.0, st[1] / st[2] - (st[0] / st[2]) ** 2)),
               episodes=st[2], diverged=st[5])
    rec['mean_reward'] = float(rew.double().mean())
    rec['mean_abs_roll_pitch'] = float(obs[:, :, :2].double().abs().mean())
    rec['mean_abs_joint'] = float(obs[:, :, 9:].double().abs().mean())
    rec['mean_speed'] = float(obs[:, :, 3:6].double().norm(dim=-1).mean())
    # late-episode posture histogram: roll bucketed in 8 bins over [-pi, pi]
    rec['roll_hist'] = torch.histc(obs[500:, :, 0].double(), bins=8, min=-np.pi, max=np.pi).cpu().numpy() / (500.0 * n)
    out[dtype] = rec
    env._close()
  a, b = out['float64'], out['float32']
  print('f64', a)
  print('f32', b)
  assert a['episodes'] == n and b['episodes'] == n and a['diverged'] == 0 and b['diverged'] == 0
  assert abs(b['mean_return'] - a['mean_return']) <= 0.01 * abs(a['mean_return'])
  assert abs(b['std_return'] - a['std_return']) <= 0.05 * a['std_return']
  assert abs(b['mean_reward'] - a['mean_reward']) <= 0.01 * abs(a['mean_reward'])
  for key in ('mean_abs_roll_pitch', 'mean_abs_joint', 'mean_speed'):
    assert abs(b[key] - a[key]) <= 0.01 * abs(a[key]), key
  assert 0.5 * np.abs(a['roll_hist'] - b['roll_hist']).sum() <= 0.01
