"""Shared builders for tests (host side only; no GPU needed to import)."""
import numpy as np

from gym_solo_amd import abi
from gym_solo_amd.core.configs import config_to_abi
from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig
from gym_solo_amd.model import JOINT_NAMES, Solo8Model


def make_abi(dtype='float64', normalize_actions=False, **kw):
  cfg = Solo8VanillaConfig()
  cfg.dtype = dtype
  for k, v in kw.items():
    setattr(cfg, k, v)
  ca = config_to_abi(cfg, cfg.starting_joint_pos, JOINT_NAMES, normalize_actions)
  return ca, Solo8Model().to_abi()


def random_actions(rng, n, scale=2 * np.pi):
  return rng.uniform(-scale, scale, (n, abi.NUM_JOINTS))


def incline_terrain(deg=10.0, n=64, cell=0.05):
  """BASELINE configs[4] (i): a plane inclined about the y axis, as a 64x64 heightfield."""
  from gym_solo_amd import abi
  xs = (np.arange(n) - 0.5 * (n - 1)) * cell
  h = np.tile(np.tan(np.radians(deg)) * xs, (n, 1))
  return abi.make_terrain(h, cell)


def stairs_terrain(rise=0.03, run=0.30, n=64, cell=0.05):
  """BASELINE configs[4] (ii): stairs climbing along +x (0.03 m rise / 0.30 m run)."""
  from gym_solo_amd import abi
  xs = (np.arange(n) - 0.5 * (n - 1)) * cell
  h = np.tile(rise * np.floor(xs / run + 0.5), (n, 1))
  return abi.make_terrain(h, cell)


def bumpy_terrain(seed=0, n=64, cell=0.05, amp=0.02):
  from gym_solo_amd import abi
  rng = np.random.default_rng(seed)
  return abi.make_terrain(amp * rng.standard_normal((n, n)), cell)


def trench_terrain(half=0.04, slope=4.0, n=96, cell=0.01):
  """A trench along x narrower than the base box: its walls touch the TOP corner spheres of the
  base while the floor side touches the bottom ones, so more than 12 of the 16 collision spheres
  can be in contact at once (impossible on a plane)."""
  from gym_solo_amd import abi
  ys = (np.arange(n) - 0.5 * (n - 1)) * cell
  h = np.clip((np.abs(ys) - half) * slope, 0.0, 0.5)
  return abi.make_terrain(np.tile(h[:, None], (1, n)), cell)


def joint_limit_case(ph, n=4, seed=0):
  """States that drive joints into their URDF limits (+-10 rad, the reference's getJointInfo fixture,
  gym_solo/core/test_obs_observations.py:123-162 columns 8-9): a settled robot lifted into the air
  with some joints placed 0.02 ... 0.2 rad inside a limit and moving towards it at 5 ... 60 rad/s, the
  motors commanding a target BEYOND the limit (12 rad, as a caller with a mutated max_motor_rotation
  can, test_solo8v2vanilla.py:110).  Returns (state [n, 32], targets [n, 12])."""
  from gym_solo_amd import abi
  rng = np.random.default_rng(seed)
  st = np.tile(ph.settle(1), (n, 1))
  st[:, abi.S_POS + 2] = 0.6
  tg = np.zeros((n, abi.NUM_JOINTS))
  for e in range(n):
    for d in range(abi.NUM_DOF):
      if rng.random() < 0.6:
        s = 1.0 if rng.random() < 0.5 else -1.0
        st[e, abi.S_Q + d] = s * (10.0 - rng.uniform(0.02, 0.2))
        st[e, abi.S_QD + d] = s * rng.uniform(5.0, 60.0)
        tg[e, 3 * (d // 2) + d % 2] = s * 12.0
  return st, tg
