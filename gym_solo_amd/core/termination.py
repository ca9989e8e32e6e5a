"""Batched termination conditions — counterpart of gym_solo/core/termination.py.

Same classes, names and error behaviour.  ``is_terminated()`` returns a ``[N]`` bool tensor
when the termination is attached to an engine (one counter per env, kept on the device and
ticked inside the fused step kernel), and keeps the reference's scalar semantics when used
stand-alone (the reference's unit tests: test_termination_conditions.py:4-38).
"""
from abc import ABC, abstractmethod

from gym_solo_amd import abi


class Termination(ABC):
  @abstractmethod
  def reset(self):
    """Resets the state of the termination condition"""
    pass

  @abstractmethod
  def is_terminated(self):
    """Determines when an episode should terminate"""
    pass

  def program(self):
    """(kind, param) for the fused kernel, or None if this termination is Python-only."""
    return None


class TerminationFactory:
  def __init__(self):
    """termination.py:19-26"""
    self._terminations = []
    self._use_or = True
    self._engine_env = None  # set by the env: enables the fused, per-env path

  def register_termination(self, *terminations):
    """termination.py:28-36"""
    self._terminations.extend(terminations)
    if self._engine_env is not None:
      self._engine_env._mark_dirty()

  def fusable(self):
    return (0 < len(self._terminations) <= abi.MAX_TERMS
            and all(t.program() is not None for t in self._terminations))

  def program(self):
    return [t.program() for t in self._terminations]

  def is_terminated(self):
    """OR over the registered conditions with short-circuit (termination.py:38-50).

    Stand-alone (no engine): exactly the reference's scalar loop.  Attached to an env: the
    per-env evaluation happens in the fused kernel; see Solo8VanillaEnv.step."""
    if not self._terminations:
      raise ValueError('Need to register at least one termination instance')
    if self._engine_env is not None:
      return self._engine_env._evaluate_terminations()
    for termination in self._terminations:
      if termination.is_terminated():
        return True
    return False

  def reset(self):
    """termination.py:52-56"""
    for termination in self._terminations:
      termination.reset()


class TimeBasedTermination(Termination):
  """termination.py:59-83: terminated once step_delta exceeds max_step_delta."""

  def __init__(self, max_step_delta: int):
    self.max_step_delta = max_step_delta
    self.reset()

  def reset(self):
    self.step_delta = 0

  def is_terminated(self) -> bool:
    self.step_delta += 1
    return self.step_delta > self.max_step_delta

  def program(self):
    return (abi.T_TIME, int(self.max_step_delta))


class PerpetualTermination(Termination):
  """termination.py:86-97: never terminates."""

  def reset(self):
    pass

  def is_terminated(self) -> bool:
    return False

  def program(self):
    return (abi.T_PERPETUAL, 0)
