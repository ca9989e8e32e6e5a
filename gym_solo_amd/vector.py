"""gym / gymnasium VectorEnv-style adapter.

``Solo8VanillaEnv`` keeps the reference's old-gym single-env signature with a leading batch axis
(``step -> (obs, reward, done, info)``).  RL libraries that consume vector envs expect
``reset() -> (obs, info)`` and ``step() -> (obs, reward, terminated, truncated, info)`` with
automatic resets; this adapter provides that on top of the in-kernel auto-reset, zero-copy (the
returned tensors alias the engine's buffers).
"""
import numpy as np

from gym_solo_amd import spaces
from gym_solo_amd.core import termination as terms


class Solo8VectorEnv:
  """Wraps a ``Solo8VanillaEnv`` created with ``config.auto_reset = True``."""

  def __init__(self, env):
    if not env.config.auto_reset:
      raise ValueError('Solo8VectorEnv needs config.auto_reset = True (in-kernel auto-reset)')
    self.env = env
    self.num_envs = env.num_envs
    self.is_vector_env = True

  @property
  def single_observation_space(self):
    return self.env.observation_space

  @property
  def single_action_space(self):
    return self.env.action_space

  @staticmethod
  def _batch(space, n):
    return spaces.Box(low=np.tile(space.low, (n, 1)), high=np.tile(space.high, (n, 1)))

  @property
  def observation_space(self):
    return self._batch(self.single_observation_space, self.num_envs)

  @property
  def action_space(self):
    return self._batch(self.single_action_space, self.num_envs)

  def _time_limited(self):
    t = self.env.termination_factory._terminations
    return bool(t) and all(isinstance(x, (terms.TimeBasedTermination, terms.PerpetualTermination)) for x in t)

  def reset(self, seed=None, options=None):
    if seed is not None:
      self.env._seed(seed)
    return self.env.reset(), {}

  def step(self, actions):
    obs, reward, done, info = self.env.step(actions)
    import torch
    never = torch.zeros_like(done) if hasattr(done, 'dtype') else False
    # a TimeBasedTermination is a time limit (truncation); anything else ends the episode
    if self._time_limited():
      return obs, reward, never, done, info
    return obs, reward, done, never, info

  def close(self):
    self.env._close()
