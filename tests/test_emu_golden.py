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
