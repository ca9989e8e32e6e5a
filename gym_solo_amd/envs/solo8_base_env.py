"""Batched Solo8 abstract base environment — counterpart of gym_solo/envs/solo8_base_env.py.

Same constructor flow (client -> gravity -> fixed timestep -> client_configuration hook ->
plane -> load_bodies -> three factories -> reset(init_call=True), solo8_base_env.py:22-55),
but ``self.client`` is a ``BatchedBulletClient`` over the HIP engine and every quantity carries
a leading env axis of size ``config.num_envs``.
"""
from abc import ABC, abstractmethod
from typing import Any, Dict, List, Tuple
import ctypes
import random

import numpy as np

from gym_solo_amd import abi, spaces
from gym_solo_amd import client as bc
from gym_solo_amd.core import configs
from gym_solo_amd.core import obs
from gym_solo_amd.core import rewards
from gym_solo_amd.core import termination as terms
import gym_solo_amd.solo_types as solo_types


class Solo8BaseEnv(ABC, spaces.Env):
  """Solo 8 abstract base environment (batched)."""
  metadata = {'render.modes': ['rgb_array']}

  def __init__(self, config: configs.Solo8BaseConfig, use_gui: bool,
               normalize_observations: bool = False):
    self.config = config
    if use_gui:
      raise ValueError('the batched MI355X engine has no GUI (use_gui must be False)')

    self.solo_model = self.build_model()
    self.engine = self.create_engine()
    self.client = bc.BatchedBulletClient(self.engine, self.solo_model,
                                         connection_mode=bc.DIRECT)
    self.client.setAdditionalSearchPath(None)
    self.client.setGravity(*self.config.gravity)

    if self.config.dt:
      self.client.setPhysicsEngineParameter(fixedTimeStep=self.config.dt, numSubSteps=1)
    else:
      self.client.setRealTimeSimulation(1)

    self.client_configuration()

    self.plane = self.client.loadURDF('plane.urdf')
    self.load_bodies()

    self.obs_factory = obs.ObservationFactory(self.client, normalize=normalize_observations)
    self.reward_factory = rewards.RewardFactory(self.client)
    self.termination_factory = terms.TerminationFactory()
    for f in (self.obs_factory, self.reward_factory, self.termination_factory):
      f._engine_env = self
    self._dirty = True
    self._copy_outputs = getattr(self, '_copy', True)
    self._fused = dict(obs=False, reward=False, done=False)
    self._valid = dict(obs=-1, reward=-1)
    self._done_from_step = False  # the last step()'s launch already evaluated the terminations

    self.reset(init_call=True)

  # ---- hooks the subclass provides ----------------------------------------------------------
  @abstractmethod
  def build_model(self):
    """Return the gym_solo_amd.model.Solo8Model to simulate."""
    pass

  @abstractmethod
  def create_engine(self):
    """Create the HIP engine (counterpart of BulletClient(connection_mode=...))."""
    pass

  @abstractmethod
  def load_bodies(self):
    """Load the bodies into the environment (solo8_base_env.py:57-65)."""
    pass

  @property
  @abstractmethod
  def action_space(self):
    pass

  @abstractmethod
  def reset(self, init_call: bool = False):
    pass

  @abstractmethod
  def step(self, action) -> Tuple[solo_types.obs, Any, Any, Dict[Any, Any]]:
    pass

  @property
  def num_envs(self) -> int:
    return self.engine.num_envs

  @property
  def observation_space(self):
    """solo8_base_env.py:107-114"""
    return self.obs_factory.get_observation_space()

  def render(self, mode='rgb_array'):
    """solo8_base_env.py:116-142 renders a pybullet camera image; out of scope here."""
    raise NotImplementedError('rendering is out of scope for the batched engine')

  def client_configuration(self):
    """Overridable hook to touch the client at init time (solo8_base_env.py:144-148)."""
    pass

  def _close(self):
    """Soft shutdown the environment (solo8_base_env.py:150-152)."""
    self.client.disconnect()

  close = _close

  def _seed(self, seed: int) -> None:
    """Seeds numpy and random (solo8_base_env.py:154-161); physics is deterministic."""
    np.random.seed(seed)
    random.seed(seed)

  # ---- fused program management ---------------------------------------------------------------
  def _mark_dirty(self):
    self._dirty = True

  def _ensure_program(self):
    if not self._dirty:
      return
    of, rf, tf = self.obs_factory, self.reward_factory, self.termination_factory
    fused = dict(obs=of.fusable(), reward=rf.fusable(), done=tf.fusable())
    prog = abi.SoloProgram()
    if fused['obs']:
      elems = of.program()
      prog.num_obs = len(elems)
      for k, e in enumerate(elems):
        o = prog.obs[k]
        o.src = e['src']
        o.flags = (abi.OBS_CLIP if e['clip'] else 0) | (abi.OBS_NORMALIZE if e['normalize'] else 0)
        o.scale, o.lo, o.hi, o.nlo, o.nhi = e['scale'], e['lo'], e['hi'], e['nlo'], e['nhi']
    if fused['reward']:
      instrs = rf.program()
      prog.num_reward_ops = len(instrs)
      for k, (op, a, b, c) in enumerate(instrs):
        r = prog.reward[k]
        r.op, r.a, r.b, r.c = op, a, b, c
    if fused['done']:
      ts = tf.program()
      prog.num_terms = len(ts)
      for k, (kind, param) in enumerate(ts):
        prog.term_kind[k] = kind
        prog.term_param[k] = param
    self.engine.set_program(prog)
    self._labels = of.labels if of._observations else []
    self._fused = fused
    self._valid = dict(obs=-1, reward=-1)
    self._done_from_step = False
    self._dirty = False

  def _flags(self, physics):
    f = abi.STEP_PHYSICS if physics else 0
    f |= abi.STEP_OBS if self._fused['obs'] else 0
    f |= abi.STEP_REWARD if self._fused['reward'] else 0
    f |= abi.STEP_DONE if self._fused['done'] else 0
    return f

  # pull-style evaluation of ONE factory on the current state (reference semantics of calling
  # get_obs / get_reward / is_terminated directly)
  def _evaluate_observations(self):
    self._ensure_program()
    if not self._fused['obs']:
      return self.obs_factory.get_obs_python()
    if self._valid['obs'] != self.client.state_version:
      self.engine.step(None, abi.STEP_OBS)
      self._valid['obs'] = self.client.state_version
    return self.engine.obs.clone() if self._copy_outputs else self.engine.obs

  def _evaluate_reward(self):
    self._ensure_program()
    if not self._fused['reward']:
      return self.reward_factory.get_reward_python()
    if self._valid['reward'] != self.client.state_version:
      self.engine.step(None, abi.STEP_REWARD)
      self._valid['reward'] = self.client.state_version
    return self.engine.reward.clone() if self._copy_outputs else self.engine.reward

  def _evaluate_terminations(self):
    self._ensure_program()
    if not self._fused['done']:
      for termination in self.termination_factory._terminations:
        if termination.is_terminated():
          return True
      return False
    if self._done_from_step:
      # step()'s own launch ticked the counters for this step: hand its result out once
      self._done_from_step = False
    else:
      # a direct TerminationFactory.is_terminated() call: every call ticks the stateful
      # terminations, as in the reference (termination.py:46-48,81-83).  A query-only launch
      # never auto-resets (the kernel ties that to SOLO_STEP_PHYSICS / SOLO_STEP_AUTO_RESET).
      self.engine.step(None, abi.STEP_DONE)
    return self.engine.done.bool()
