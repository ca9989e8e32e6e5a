"""Batched Solo8VanillaEnv — counterpart of gym_solo/envs/solo8v2vanilla.py."""
from dataclasses import dataclass

import numpy as np

from gym_solo_amd.core.configs import Solo8BaseConfig


@dataclass
class Solo8VanillaConfig(Solo8BaseConfig):
  """gym_solo/envs/solo8v2vanilla.py:18-34 (same defaults; ``starting_joint_pos`` is a plain
  class attribute there too)."""
  urdf_path: str = 'assets/solo8v2/solo.urdf'
  starting_joint_pos = {
    'FL_HFE': np.pi / 2,
    'FL_KFE': np.pi,
    'FL_ANKLE': 0,
    'FR_HFE': np.pi / 2,
    'FR_KFE': np.pi,
    'FR_ANKLE': 0,
    'HL_HFE': -np.pi / 2,
    'HL_KFE': -np.pi,
    'HL_ANKLE': 0,
    'HR_HFE': -np.pi / 2,
    'HR_KFE': -np.pi,
    'HR_ANKLE': 0
  }


from typing import Any, Dict, List, Tuple  # noqa: E402
import time  # noqa: E402

from gym_solo_amd import abi, spaces, solo_types  # noqa: E402
from gym_solo_amd import client as p  # noqa: E402  (plays the role of `import pybullet as p`)
from gym_solo_amd.core.configs import config_to_abi  # noqa: E402
from gym_solo_amd.envs.solo8_base_env import Solo8BaseEnv  # noqa: E402
from gym_solo_amd.model import JOINT_NAMES, Solo8Model  # noqa: E402


class Solo8VanillaEnv(Solo8BaseEnv):
  """The unmodified solo8 gym environment, batched over ``config.num_envs`` robots that are
  stepped by one fused HIP kernel launch (gym_solo/envs/solo8v2vanilla.py:37-172)."""

  def __init__(self, use_gui: bool = False, realtime: bool = False, config=None,
               normalize_actions: bool = False, normalize_observations: bool = False,
               copy_outputs: bool = True, **kwargs):
    self._realtime = realtime
    self._normalize = normalize_actions
    self._copy = copy_outputs
    super().__init__(config or Solo8VanillaConfig(), use_gui,
                     normalize_observations=normalize_observations)

  def build_model(self):
    """loadURDF(self.config.urdf, ...) (solo8v2vanilla.py:151-155) when the file exists, else the
    built-in solo8v2 constants."""
    if self.config.urdf:
      from gym_solo_amd.urdf import load_urdf
      return load_urdf(self.config.urdf)
    return Solo8Model()

  def create_engine(self):
    from gym_solo_amd.engine import Engine
    cfg = config_to_abi(self.config, self.config.starting_joint_pos, JOINT_NAMES,
                        normalize_actions=self._normalize)
    engine = Engine(cfg, self.solo_model.to_abi(), self.config.num_envs, self.config.device)
    terrain = getattr(self.config, 'terrain', None)
    if terrain is not None:
      # loadURDF('plane.urdf') -> heightfield (BASELINE configs[4]); re-settles on the new ground
      if isinstance(terrain, dict):
        terrain = abi.make_terrain(terrain['heights'], terrain['cell'], terrain.get('origin'))
      engine.set_terrain(terrain)
    return engine

  @property
  def action_space(self):
    """solo8v2vanilla.py:51-70"""
    if not self._action_space:
      raise ValueError('No valid action space')

    if self._normalize:
      return spaces.Box(low=-1, high=1, shape=self._action_space.shape)
    else:
      return self._action_space

  def step(self, action) -> Tuple[solo_types.obs, Any, Any, Dict[Any, Any]]:
    """One env step for all robots (solo8v2vanilla.py:72-102).

    action: ``[N, 12]`` tensor (or a 12-vector applied to every robot) of joint position
    targets; de-normalisation (:84-85) happens inside the kernel.  Returns ``(obs [N,D],
    reward [N], done [N] bool, {'labels': ...})``.
    """
    # same failure order as the reference: get_obs, get_reward, is_terminated (:96-100)
    if not self.obs_factory._observations:
      raise ValueError('Need to register at least one observation instance')
    if not self.reward_factory._rewards:
      raise ValueError('Need to register at least one reward instance')
    if not self.termination_factory._terminations:
      raise ValueError('Need to register at least one termination instance')
    self._ensure_program()
    eng = self.engine
    actions = self.client.as_actions(action)
    fused = self._fused
    if fused['obs'] and fused['reward'] and fused['done']:
      # setJointMotorControlArray + stepSimulation + obs/reward/done reductions: ONE launch
      eng.step(actions, self._flags(physics=True))
      self.client.state_version += 1
      v = self.client.state_version
      if not self.config.auto_reset:  # (after an in-kernel auto-reset the buffers hold the terminal outputs,
        self._valid['obs'] = self._valid['reward'] = v  # not those of the restored state: a later pull re-evaluates)
      if self._realtime:
        time.sleep(self.config.dt)
      # everything was produced by that launch: hand the engine's buffers out without going
      # through the three pull-style factory calls (host time per step matters in closed loop)
      if self._copy_outputs:
        return eng.obs.clone(), eng.reward.clone(), eng.done.bool(), {'labels': self._labels}
      return eng.obs, eng.reward, eng.done_bool, {'labels': self._labels}

    # At least one factory holds a Python-only member (a custom Observation / Reward /
    # Termination without program()): the launch advances the physics and evaluates what is
    # fusable; the Python members read the post-step state through the client; the terminations
    # - and with them the auto-reset - come LAST, so that no Python member ever sees a state the
    # kernel has already reset (reference order: get_obs, get_reward, is_terminated, :96-100).
    eng.step(actions, self._flags(physics=True) & ~abi.STEP_DONE)
    self.client.state_version += 1
    v = self.client.state_version
    for key in ('obs', 'reward'):
      if fused[key]:
        self._valid[key] = v

    if self._realtime:
      time.sleep(self.config.dt)

    obs_values, obs_labels = self.obs_factory.get_obs()
    reward = self.reward_factory.get_reward()
    if fused['done']:
      eng.step(None, abi.STEP_DONE | abi.STEP_AUTO_RESET)
      self._done_from_step = True
    done = self.termination_factory.is_terminated()
    if self.config.auto_reset:
      if fused['done']:
        self.client.state_version += 1  # (the launch above restored the finished robots)
      elif done is True or (hasattr(done, 'any') and bool(done.any())):
        self.reset_where(done)
        if done is True:
          # host-side terminations are per-object, not per-robot: a scalar True ended the episode of the
          # whole batch, so their per-episode state restarts with it (reset(), solo8v2vanilla.py:110-143;
          # left alone, a TimeBasedTermination stays past its limit and the batch is reset every step).
          # A per-robot flag tensor comes from a custom termination that keeps per-robot state itself.
          self.termination_factory.reset()
    return obs_values, reward, done, {'labels': obs_labels}

  def reset_where(self, done):
    """Restore the robots whose flag is set (``True`` = all) without touching the host-side
    termination objects' per-episode state of the others."""
    import torch
    if done is True:
      self.engine.reset(None)
    else:
      self.engine.reset(torch.as_tensor(done).to(device=self.engine.state.device, dtype=torch.uint8).contiguous())
    self.client.state_version += 1

  def reset(self, init_call: bool = False, mask=None):
    """Restore the post-settle snapshot (solo8v2vanilla.py:104-143).  The reference rebuilds
    the world and re-runs the 500 settle steps; that loop is deterministic, so the engine runs
    it once at creation and reset copies the result.  ``mask`` ([N] uint8/bool tensor) limits
    the reset to some robots (build extension)."""
    if mask is not None:
      import torch
      mask = mask.to(device=self.engine.state.device, dtype=torch.uint8).contiguous()
    self.engine.reset(mask)
    self.client.state_version += 1
    self.client.setGravity(*self.config.gravity)
    if self.config.dt:
      self.client.setPhysicsEngineParameter(fixedTimeStep=self.config.dt, numSubSteps=1)
    self.client_configuration()
    self.termination_factory.reset()

    if init_call:
      return np.empty(shape=(0,)), []
    else:
      obs_values, _ = self.obs_factory.get_obs()
      return obs_values

  def load_bodies(self):
    """Counterpart of solo8v2vanilla.py:145-172 on the batched client: the robot at the configured
    start pose with inertias from the file, the four dynamics parameters applied to every joint's
    link, and the 12-joint action box.  (The facade validates these calls against the engine's
    compiled configuration; a differing lateral friction becomes the per-env parameter row.)"""
    cfg, client = self.config, self.client
    self.robot = client.loadURDF('solo8v2/solo.urdf', cfg.robot_start_pos,
                                 client.getQuaternionFromEuler(cfg.robot_start_orientation_euler),
                                 flags=p.URDF_USE_INERTIA_FROM_FILE, useFixedBase=False)
    self._joint_cnt = client.getNumJoints(self.robot)
    dynamics = dict(linearDamping=cfg.linear_damping, angularDamping=cfg.angular_damping,
                    restitution=cfg.restitution, lateralFriction=cfg.lateral_friction)
    self.joint_ordering = []
    for joint in range(self._joint_cnt):
      client.changeDynamics(self.robot, joint, **dynamics)
      self.joint_ordering.append(client.getJointInfo(self.robot, joint)[1].decode('UTF-8'))
    self._zero_gains = np.zeros(self._joint_cnt)
    self._action_space = spaces.Box(-cfg.max_motor_rotation, cfg.max_motor_rotation,
                                    shape=(self._joint_cnt,))
