"""Host-side API (envs, factories, client facade) on the CPU, backed by the fibre emulator
running the product kernel source (tests/emu_kernel.py) — no GPU needed."""
import numpy as np
import pytest

import env_cases as cases
from emu_kernel import make_emu_env_class
from gym_solo_amd.core.configs import config_to_abi

EmuSolo8VanillaEnv = make_emu_env_class()


def make_env(config=None, **kw):
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig
  config = config or Solo8VanillaConfig()
  if not getattr(config, '_dtype_pinned', False):
    config.dtype = 'float64'
  if not getattr(config, '_num_envs_pinned', False):
    config.num_envs = getattr(make_env, 'num_envs', 2)
  return EmuSolo8VanillaEnv(config=config, **kw)


def test_action_space():
  cases.case_action_space(make_env)


def test_step_no_rewards():
  cases.case_step_no_rewards(make_env)


def test_step_simple_reward():
  cases.case_step_simple_reward(make_env)


def test_action_normalization():
  cases.case_action_normalization(make_env)


def test_action_normalization_float32_bound():
  cases.case_action_normalization_float32_bound(make_env)


def test_reset():
  cases.case_reset(make_env)


def test_actions_rest_and_motion():
  make_env.num_envs = 1
  try:
    cases.case_actions_rest_and_motion(make_env)
  finally:
    make_env.num_envs = 2


def test_disjoint_environments():
  cases.case_disjoint_environments(make_env)


@pytest.mark.parametrize('normalize', [False, True])
def test_fused_matches_python_and_oracle(normalize):
  cases.case_fused_matches_python_and_oracle(make_env, steps=12, tol=1e-9,
                                             normalize_observations=normalize)


def test_partial_fused_auto_reset():
  cases.case_partial_fused_auto_reset(make_env)


def test_reset_restores_motor_targets():
  cases.case_reset_restores_motor_targets(make_env)


def test_gui_and_realtime_flags():
  with pytest.raises(ValueError):
    make_env(use_gui=True)
  from unittest import mock
  env = make_env(realtime=True)
  from gym_solo_amd.testing import CompliantObs, DummyTermination, SimpleReward
  env.reward_factory.register_reward(1, SimpleReward())
  env.obs_factory.register_observation(CompliantObs(None))
  env.termination_factory.register_termination(DummyTermination(0, True))
  with mock.patch('time.sleep', return_value=None) as sl:
    env.step(env.action_space.sample())
    assert sl.called  # gym_solo/envs/test_solo8v2vanilla.py:37-48


def test_make_and_vector_adapter(monkeypatch):
  """gym_solo/__init__.py:3-11 ids + a VectorEnv-style adapter over the in-kernel auto-reset."""
  import torch
  import gym_solo_amd
  from gym_solo_amd.envs import solo8v2vanilla
  from gym_solo_amd.vector import Solo8VectorEnv
  from gym_solo_amd.core import obs as solo_obs, termination as terms
  from gym_solo_amd.testing import SimpleReward
  with pytest.raises(ValueError):
    gym_solo_amd.make('solo8vanilla-realtime-v0')
  with pytest.raises(ValueError):
    gym_solo_amd.make('nope-v0')
  monkeypatch.setitem(gym_solo_amd._REGISTRY, 'solo8vanilla-v0', 'test_env_host:EmuSolo8VanillaEnv')
  cfg = solo8v2vanilla.Solo8VanillaConfig()
  cfg.dtype, cfg.num_envs, cfg.auto_reset = 'float64', 2, True
  env = gym_solo_amd.make('solo8vanilla-v0', config=cfg)
  env.obs_factory.register_observation(solo_obs.TorsoIMU(env.robot))
  env.reward_factory.register_reward(1, SimpleReward())
  env.termination_factory.register_termination(terms.TimeBasedTermination(2))
  venv = Solo8VectorEnv(env)
  assert venv.num_envs == 2 and venv.observation_space.shape == (2, 9) and venv.action_space.shape == (2, 12)
  obs0, info = venv.reset(seed=3)
  assert obs0.shape == (2, 9) and info == {}
  flags = []
  for k in range(6):
    o, r, terminated, truncated, info = venv.step(torch.zeros(2, 12, dtype=torch.float64))
    flags.append((bool(terminated.any()), bool(truncated.all())))
  assert flags == [(False, False), (False, False), (False, True)] * 2  # time limit -> truncation
  cfg2 = solo8v2vanilla.Solo8VanillaConfig()
  cfg2.dtype, cfg2.num_envs = 'float64', 2
  with pytest.raises(ValueError):
    Solo8VectorEnv(make_env(config=cfg2))


def test_host_termination_with_auto_reset_restarts_the_episode():
  cases.case_host_termination_auto_reset(make_env)


def test_residual_threshold_is_validated():
  """SoloConfig.solver_residual_threshold (pybullet's solverResidualThreshold, opt-in): 0 by default, negative and NaN
  values are rejected on the host like the other solver knobs."""
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig
  from gym_solo_amd.model import JOINT_NAMES
  cfg = Solo8VanillaConfig()
  assert cfg.solver_residual_threshold == 0.0
  assert config_to_abi(cfg, cfg.starting_joint_pos, JOINT_NAMES).solver_residual_threshold == 0.0
  cfg.solver_residual_threshold = 1e-7
  assert config_to_abi(cfg, cfg.starting_joint_pos, JOINT_NAMES).solver_residual_threshold == 1e-7
  for bad in (-1e-9, float('nan')):
    cfg.solver_residual_threshold = bad
    with pytest.raises(ValueError):
      config_to_abi(cfg, cfg.starting_joint_pos, JOINT_NAMES)
