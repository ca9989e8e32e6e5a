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
