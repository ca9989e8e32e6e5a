"""Batched observations — counterpart of gym_solo/core/obs.py.

Same classes (``Observation``, ``ObservationFactory``, ``TorsoIMU``, ``MotorEncoder``), argument
names, labels, spaces and ``ValueError`` sites.  Two evaluation paths:

* fused — every registered observation describes itself as a list of ``SoloObsElem`` (source
  index, scale, clip and normalisation bounds); the factory's list is compiled into the engine's
  program and evaluated inside the step kernel (lane k computes element k).
* Python — ``compute()`` pulls state through the batched client's getters (torch ops on the
  GPU tensors), exactly like the reference pulls through pybullet.  Used for custom
  observations and as an independent check of the fused path in the tests.
"""
from abc import ABC, abstractmethod
from typing import List, Tuple

import numpy as np

from gym_solo_amd import abi, solo_types, spaces


def _is_tensor(x):
  return type(x).__module__.startswith('torch')


class Observation(ABC):
  """An observation for a body in the simulation (obs.py:16-92)."""
  _client = None

  @abstractmethod
  def __init__(self, body_id: int):
    pass

  @property
  @abstractmethod
  def observation_space(self):
    pass

  @property
  @abstractmethod
  def labels(self) -> List[str]:
    pass

  @abstractmethod
  def compute(self) -> solo_types.obs:
    pass

  @property
  def client(self):
    if not self._client:
      raise ValueError('PyBullet client needs to be set')
    return self._client

  @client.setter
  def client(self, client):
    self._client = client

  def program(self):
    """List of abi.SoloObsElem-like dicts (src, scale, clip, lo, hi) for the fused kernel, or
    None when the observation can only be computed in Python."""
    return None


class ObservationFactory:
  def __init__(self, client, normalize: bool = False):
    """obs.py:96-107"""
    self._client = client
    self._observations = []
    self._obs_space = None
    self._normalize = normalize
    self._engine_env = None

  def register_observation(self, obs: Observation):
    """obs.py:109-128: sets the client, then validates label / space / value lengths."""
    obs.client = self._client

    lbl_len = len(obs.labels)
    obs_space_len = len(obs.observation_space.low)
    value = obs.compute()
    obs_len = int(value.shape[-1]) if _is_tensor(value) else int(np.asarray(value).shape[-1])

    if lbl_len != obs_space_len:
      raise ValueError('Labels have length {} != obs space len {}'.format(lbl_len, obs_space_len))
    if lbl_len != obs_len:
      raise ValueError('Labels have length {} != obs len {}'.format(lbl_len, obs_len))

    self._observations.append(obs)
    if self._engine_env is not None:
      self._engine_env._mark_dirty()

  # ---- fused path -----------------------------------------------------------------------
  def fusable(self):
    if not self._observations:
      return False
    progs = [o.program() for o in self._observations]
    return all(p is not None for p in progs) and sum(len(p) for p in progs) <= abi.MAX_OBS

  def program(self) -> List[dict]:
    elems = []
    for o in self._observations:
      space = o.observation_space
      for k, e in enumerate(o.program()):
        e = dict(e)
        e['normalize'] = bool(self._normalize)
        # float32 Box bounds, float64 arithmetic (obs.py:149-152)
        e['nlo'] = float(np.float32(space.low[k]))
        e['nhi'] = float(np.float32(space.high[k]))
        elems.append(e)
    return elems

  @property
  def labels(self):
    return [l for o in self._observations for l in o.labels]

  # ---- Python path (reference semantics) --------------------------------------------------
  def get_obs_python(self):
    all_obs = []
    for obs in self._observations:
      values = obs.compute()
      if _is_tensor(values):
        import torch
        if self._normalize:
          low = torch.as_tensor(np.asarray(obs.observation_space.low, dtype=np.float32),
                                device=values.device).to(values.dtype)
          hi = torch.as_tensor(np.asarray(obs.observation_space.high, dtype=np.float32),
                               device=values.device).to(values.dtype)
          values = ((2 * (values - low)) / (hi - low)) - 1
      else:
        values = np.asarray(values)
        if self._normalize:
          low = obs.observation_space.low
          hi = obs.observation_space.high
          values = ((2 * (values - low)) / (hi - low)) - 1
      all_obs.append(values)
    if any(_is_tensor(v) for v in all_obs):
      import torch
      ref = next(v for v in all_obs if _is_tensor(v))
      parts = []
      for v in all_obs:
        if not _is_tensor(v):
          v = torch.as_tensor(np.asarray(v), device=ref.device).to(ref.dtype)
        if v.dim() == 1:
          v = v.unsqueeze(0).expand(ref.shape[0], -1)
        parts.append(v)
      return torch.cat(parts, dim=-1)
    return np.concatenate(all_obs, axis=-1)

  def get_obs(self) -> Tuple[solo_types.obs, List[str]]:
    """obs.py:130-159: all observations of the current state, concatenated, plus labels."""
    if not self._observations:
      raise ValueError('Need to register at least one observation instance')
    if self._engine_env is not None:
      return self._engine_env._evaluate_observations(), self.labels
    return self.get_obs_python(), self.labels

  def get_observation_space(self, generate=False):
    """obs.py:161-190"""
    if not self._observations:
      raise ValueError('Can\'t generate an empty observation space')
    if self._obs_space and not generate:
      return self._obs_space

    lower, upper = [], []
    for obs in self._observations:
      lower.extend(obs.observation_space.low)
      upper.extend(obs.observation_space.high)

    if self._normalize:
      self._obs_space = spaces.Box(low=-1, high=1, shape=(len(lower),))
    else:
      self._obs_space = spaces.Box(low=np.array(lower), high=np.array(upper))
    return self._obs_space


class TorsoIMU(Observation):
  """Orientation and velocities of the Solo 8 torso (obs.py:193-282)."""
  labels: List[str] = ['θx', 'θy', 'θz', 'vx', 'vy', 'vz', 'wx', 'wy', 'wz']

  def __init__(self, body_id: int, degrees: bool = False, max_lin_velocity: float = 15,
               max_angular_velocity: float = 10.):
    self.robot = body_id
    self._degrees = degrees
    self._max_lin = max_lin_velocity
    self._max_ang = max_angular_velocity

    self._low = None
    self._high = None
    self.observation_space  # Populate the bounds in case it doesn't get called

  @property
  def observation_space(self):
    """obs.py:228-258 (bounds are frozen on first access, :254-256)."""
    angle_min = -180. if self._degrees else -np.pi
    angle_max = 180. if self._degrees else np.pi

    lower = [angle_min, angle_min, angle_min,
             -self._max_lin, -self._max_lin, -self._max_lin,
             -self._max_ang, -self._max_ang, -self._max_ang]
    upper = [angle_max, angle_max, angle_max,
             self._max_lin, self._max_lin, self._max_lin,
             self._max_ang, self._max_ang, self._max_ang]

    if not (self._low and self._high):
      self._low = lower
      self._high = upper

    return spaces.Box(low=np.array(lower), high=np.array(upper))

  def compute(self) -> solo_types.obs:
    """obs.py:260-282, batched: [N, 9] (or [9] with a scalar mock client)."""
    _, orien_quat = self.client.getBasePositionAndOrientation(self.robot)
    orien = self.client.getEulerFromQuaternion(orien_quat)
    v_lin, v_ang = self.client.getBaseVelocity(self.robot)
    if _is_tensor(orien):
      import torch
      if self._degrees:  # angles and angular velocity only; v_lin stays (obs.py:277-279)
        orien = torch.rad2deg(orien)
        v_ang = torch.rad2deg(v_ang)
      raw = torch.cat([orien, v_lin, v_ang], dim=-1)
      low = torch.as_tensor(self._low, device=raw.device, dtype=raw.dtype)
      high = torch.as_tensor(self._high, device=raw.device, dtype=raw.dtype)
      return torch.minimum(torch.maximum(raw, low), high)
    orien = np.array(orien)
    v_lin = np.array(v_lin)
    v_ang = np.array(v_ang)
    if self._degrees:
      orien = np.degrees(orien)
      v_ang = np.degrees(v_ang)
    raw_values = np.concatenate([orien, v_lin, v_ang])
    return np.clip(raw_values, self._low, self._high)

  def program(self):
    deg = 180.0 / np.pi if self._degrees else 1.0
    elems = []
    for k in range(9):
      scale = deg if (k < 3 or k >= 6) else 1.0
      elems.append(dict(src=abi.SRC_EULER + k, scale=scale, clip=True,
                        lo=float(self._low[k]), hi=float(self._high[k])))
    return elems


class MotorEncoder(Observation):
  """Position of all the joints (obs.py:285-363)."""

  def __init__(self, body_id: int, degrees: bool = False, max_rotation: float = None):
    self.robot = body_id
    self._degrees = degrees
    self._max_rot = max_rotation

  @property
  def _num_joints(self):
    return self.client.getNumJoints(self.robot)

  @property
  def observation_space(self):
    """obs.py:310-334"""
    if self._max_rot:
      return spaces.Box(low=-self._max_rot, high=self._max_rot, shape=(self._num_joints, ))

    lower, upper = [], []
    for joint in range(self._num_joints):
      joint_info = self.client.getJointInfo(self.robot, joint)
      lower.append(joint_info[8])
      upper.append(joint_info[9])

    lower = np.array(lower)
    upper = np.array(upper)

    if self._degrees:
      lower = np.degrees(lower)
      upper = np.degrees(upper)

    return spaces.Box(low=lower, high=upper)

  @property
  def labels(self) -> List[str]:
    """obs.py:336-345"""
    return [self.client.getJointInfo(self.robot, joint)[1].decode('UTF-8')
            for joint in range(self._num_joints)]

  def compute(self) -> solo_types.obs:
    """obs.py:347-363, batched: [N, 12]."""
    vals = [self.client.getJointState(self.robot, i)[0] for i in range(self._num_joints)]
    if any(_is_tensor(v) for v in vals):
      import torch
      joint_values = torch.stack(vals, dim=-1)
      if self._degrees:
        joint_values = torch.rad2deg(joint_values)
      if self._max_rot:
        joint_values = torch.clamp(joint_values, -self._max_rot, self._max_rot)
      return joint_values
    joint_values = np.array(vals)
    if self._degrees:
      joint_values = np.degrees(joint_values)
    if self._max_rot:
      joint_values = np.clip(joint_values, -self._max_rot, self._max_rot)
    return joint_values

  def program(self):
    deg = 180.0 / np.pi if self._degrees else 1.0
    clip = bool(self._max_rot)
    lim = float(self._max_rot) if clip else 0.0
    return [dict(src=abi.SRC_JPOS + j, scale=deg, clip=clip, lo=-lim, hi=lim)
            for j in range(abi.NUM_JOINTS)]
