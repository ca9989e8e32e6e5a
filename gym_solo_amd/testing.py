"""Test doubles — counterpart of gym_solo/testing.py:11-79 (same names and behaviour)."""
import numpy as np

from gym_solo_amd import abi, solo_types, spaces
from gym_solo_amd.core import obs
from gym_solo_amd.core import rewards
from gym_solo_amd.core import termination


class CompliantObs(obs.Observation):
  """Always observes [1, 2] in Box([0,0],[3,3]) (testing.py:11-36)."""
  observation_space = spaces.Box(low=np.array([0., 0.]), high=np.array([3., 3.]))
  labels = ['1', '2']

  def __init__(self, body_id: int):
    pass

  def compute(self) -> solo_types.obs:
    return np.array([1., 2.])

  def program(self):
    return [dict(src=abi.SRC_ONE, scale=1.0, clip=False, lo=0.0, hi=0.0),
            dict(src=abi.SRC_ONE, scale=2.0, clip=False, lo=0.0, hi=0.0)]


class SimpleReward(rewards.Reward):
  """Always 1 (testing.py:39-47)."""

  def compute(self) -> float:
    return 1

  def program(self):
    return [(abi.R_CONST, 1.0, 0.0, 0.0)]


class ReflectiveReward(rewards.Reward):
  """A configurable fixed value (testing.py:50-66)."""

  def __init__(self, return_value: float):
    self._return_value = return_value

  def compute(self) -> float:
    return self._return_value

  def program(self):
    return [(abi.R_CONST, float(self._return_value), 0.0, 0.0)]


class DummyTermination(termination.Termination):
  """testing.py:69-79"""

  def __init__(self, body_id: int, termination_var: bool):
    self.body_id = body_id
    self.termination_var = termination_var
    self.reset_counter = 0
    self.reset()

  def reset(self):
    self.reset_counter += 1

  def is_terminated(self) -> bool:
    return self.termination_var

  def program(self):
    return (abi.T_CONST, 1 if self.termination_var else 0)
