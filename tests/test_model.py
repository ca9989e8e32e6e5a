"""Model constants vs the reference's getJointInfo fixture
(gym_solo/core/test_obs_observations.py:123-162, committed as data in
tests/golden/joint_info_fixture.json)."""
import json
import os

import numpy as np

from gym_solo_amd import abi
from gym_solo_amd.model import JOINT_NAMES, Solo8Model, pybullet_joint_info

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'joint_info_fixture.json')


def test_joint_info_matches_fixture():
  fixture = json.load(open(GOLD))['joint_info']
  ours = pybullet_joint_info(Solo8Model())
  assert len(ours) == len(fixture) == 12
  for f, o in zip(fixture, ours):
    assert o[0] == f[0]
    assert o[1].decode() == f[1]            # joint name
    assert o[2] == f[2]                     # type (0 revolute / 4 fixed)
    assert (o[3], o[4], o[5]) == (f[3], f[4], f[5])  # qIndex, uIndex, flags
    assert (o[6], o[7]) == (f[6], f[7])     # damping, friction
    assert (o[8], o[9], o[10], o[11]) == (f[8], f[9], f[10], f[11])  # limits, maxForce, maxVel
    assert o[12].decode() == f[12]          # link name
    np.testing.assert_allclose(o[13], f[13], atol=2e-5)   # axis in the principal-axis frame
    np.testing.assert_allclose(o[14], f[14], atol=2e-6)   # parentFramePos
    np.testing.assert_allclose(o[15], f[15], atol=2e-5)   # parentFrameOrn
    assert o[16] == f[16]                   # parent index


def test_abi_model_is_consistent():
  m = Solo8Model()
  a = m.to_abi()
  assert a.num_spheres == 16
  assert abs(sum(a.mass) - m.total_mass) < 1e-12
  assert list(a.dof_to_joint) == [0, 1, 3, 4, 6, 7, 9, 10]
  assert [JOINT_NAMES[j] for j in a.dof_to_joint] == [
    'FL_HFE', 'FL_KFE', 'FR_HFE', 'FR_KFE', 'HL_HFE', 'HL_KFE', 'HR_HFE', 'HR_KFE']
  for j in range(abi.NUM_DOF):
    assert list(a.joint_axis[j]) == [0.0, 1.0, 0.0]
    assert a.parent[j] == (0 if j % 2 == 0 else j)
  # lower leg + welded foot: mass adds up, inertia stays positive definite
  for leg in range(4):
    li = m.lower_with_foot(leg)
    assert abs(li.mass - (m.lower_mass + m.foot_mass)) < 1e-15
    assert np.linalg.eigvalsh(li.inertia).min() > 0
