"""Reference-generated golden vectors through the PRODUCT kernel source on the CPU wave emulator
(tests/golden_cases.py); the GPU suite runs the same bodies on the HIP engine."""
import pytest

import golden_cases as gc
from test_env_host import make_env


@pytest.mark.parametrize('normalize', [False, True])
@pytest.mark.parametrize('name', sorted(gc.OBS))
def test_observations_match_reference_f64(name, normalize):
  gc.case_observations(make_env, name, 'float64', normalize)


@pytest.mark.parametrize('name', ['imu_deg', 'enc_deg_clip', 'bench'])
def test_observations_match_reference_f32(name):
  gc.case_observations(make_env, name, 'float32', True)


@pytest.mark.parametrize('name', sorted(gc.REW))
def test_rewards_match_reference_f64(name):
  gc.case_reward(make_env, name, 'float64')


@pytest.mark.parametrize('name', ['upright', 'hard_step', 'composite', 'weighted3'])
def test_rewards_match_reference_f32(name):
  gc.case_reward(make_env, name, 'float32')


def test_termination_sequences_match_reference():
  gc.case_terminations(make_env)


def test_random_reward_trees_fused_vs_reference_semantics():
  """Randomly nested Additive / Multiplicitive trees over every physical reward, compiled by the host
  factories into the postfix program and evaluated by the product kernel (emulator), against the
  pull-style Python evaluation of the same tree (reference semantics, rewards.py:104-186) on the
  golden states: the compiler (three-address form, weights, nesting, left-to-right sums) is exact."""
  import numpy as np
  import torch
  from gym_solo_amd import abi
  from gym_solo_amd.core import rewards
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig
  g = gc.gold()
  st = gc.golden_state(g)[:32]
  rng = np.random.default_rng(11)

  def leaf(env):
    r, k = env.robot, rng.integers(0, 5)
    return [lambda: rewards.UprightReward(r),
            lambda: rewards.FlatTorsoReward(r, hard_margin=float(rng.uniform(0, .3)), soft_margin=float(rng.uniform(0, 2))),
            lambda: rewards.TorsoHeightReward(r, float(rng.uniform(.1, .4)), float(rng.uniform(0, .1)), float(rng.uniform(0, .3))),
            lambda: rewards.HorizontalMoveSpeedReward(r, float(rng.uniform(0, 2)), float(rng.uniform(0, .5)), float(rng.uniform(0, 3))),
            lambda: rewards.SmallControlReward(r, margin=float(rng.uniform(0, 12)))][k]()

  def tree(env, depth):
    if depth == 0 or rng.random() < 0.3:
      return leaf(env)
    if rng.random() < 0.5:
      node = rewards.AdditiveReward()
      node.client = env.client
      for _ in range(rng.integers(1, 4)):
        node.add_term(float(rng.uniform(-2, 2)), tree(env, depth - 1))
      return node
    return rewards.MultiplicitiveReward(float(rng.uniform(-2, 2)), *[tree(env, depth - 1) for _ in range(rng.integers(1, 4))])

  done = 0
  while done < 20:
    cfg = Solo8VanillaConfig()
    cfg.dtype, cfg._dtype_pinned, cfg.num_envs, cfg._num_envs_pinned, cfg.settle_steps = 'float64', True, st.shape[0], True, 0
    env = make_env(config=cfg)
    for _ in range(rng.integers(1, 4)):
      env.reward_factory.register_reward(float(rng.uniform(-3, 3)), tree(env, 2))
    if not env.reward_factory.fusable():      # (longer than SOLO_MAX_REWARD_OPS: stays on the Python path)
      continue
    env._ensure_program()
    assert env._fused['reward']
    gc._load_state(env, st)
    env.engine.step(None, abi.STEP_REWARD)
    fused = gc._np(env.engine.reward)
    python = gc._np(env.reward_factory.get_reward_python())
    np.testing.assert_allclose(fused, python, rtol=1e-12, atol=1e-12)
    done += 1
