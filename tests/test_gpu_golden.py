"""Reference-generated golden vectors (tests/golden/obs_reward_golden.*) straight through the HIP
observation / reward / termination kernels, via the C-ABI (tests/golden_cases.py): every program in
tests/test_oracle_golden.py's OBS / REW tables, f64 and f32, plus the termination sequences."""
import pytest

import golden_cases as gc
from test_gpu_env import make_env

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
@pytest.mark.parametrize('normalize', [False, True])
@pytest.mark.parametrize('name', sorted(gc.OBS))
def test_observations_match_reference(name, normalize, dtype):
  gc.case_observations(make_env, name, dtype, normalize)


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
@pytest.mark.parametrize('name', sorted(gc.REW))
def test_rewards_match_reference(name, dtype):
  gc.case_reward(make_env, name, dtype)


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_termination_sequences_match_reference(dtype):
  gc.case_terminations(make_env, dtype)


def test_random_reward_trees_on_the_hip_engine():
  """The randomized reward-tree check of tests/test_emu_golden.py on the HIP engine (f64)."""
  import test_emu_golden as cpu
  saved = cpu.make_env
  cpu.make_env = make_env
  try:
    cpu.test_random_reward_trees_fused_vs_reference_semantics()
  finally:
    cpu.make_env = saved
