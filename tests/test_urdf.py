"""URDF loader: write the built-in constants as URDF, load them back, and check
the C-ABI model and the reference's getJointInfo fixture are reproduced from the FILE."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from gym_solo_amd import abi
from gym_solo_amd.model import Solo8Model, pybullet_joint_info
from gym_solo_amd.urdf import UrdfGeometryWarning, load_urdf, parse_urdf, to_urdf


def test_round_trip_reproduces_abi_model(tmp_path):
  path = tmp_path / 'solo.urdf'
  path.write_text(to_urdf(Solo8Model()))
  loaded = load_urdf(str(path))
  a, b = Solo8Model().to_abi(), loaded.to_abi()
  assert bytes(C.string_at(C.addressof(a), C.sizeof(a))) == bytes(C.string_at(C.addressof(b), C.sizeof(b)))
  assert loaded.total_mass == pytest.approx(Solo8Model().total_mass, rel=1e-15)


def test_fixture_reproduced_from_urdf():
  fixture = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'joint_info_fixture.json')))
  info = pybullet_joint_info(parse_urdf(to_urdf()))
  for f, o in zip(fixture['joint_info'], info):
    assert o[1].decode() == f[1] and o[2] == f[2] and o[16] == f[16]
    np.testing.assert_allclose(o[13], f[13], atol=2e-5)
    np.testing.assert_allclose(o[14], f[14], atol=2e-6)
    np.testing.assert_allclose(o[15], f[15], atol=2e-5)


def test_modified_urdf_changes_the_model_and_bad_urdfs_are_rejected():
  text = to_urdf()
  heavier = parse_urdf(text.replace('<mass value="1.16115091"/>', '<mass value="2.0"/>'))
  assert heavier.to_abi().mass[0] == 2.0
  rotated = text.replace('<origin xyz="1.377e-05 0.01935853 -0.078707" rpy="0 0 0"/>',
                         '<origin xyz="1.377e-05 0.01935853 -0.078707" rpy="0.3 0 0"/>', 1)
  m = parse_urdf(rotated).to_abi()
  assert abs(m.inertia[1][5] - Solo8Model().to_abi().inertia[1][5]) > 1e-5  # Iyz of the FL upper leg
  with pytest.raises(ValueError):
    parse_urdf(text.replace('<axis xyz="0 1 0"/>', '<axis xyz="1 0 0"/>', 1))
  with pytest.raises(ValueError):
    parse_urdf(text.replace('name="HR_ANKLE" type="fixed"', 'name="HR_ANKLE" type="revolute"'))
  with pytest.raises(ValueError):
    parse_urdf(text.replace('FL_KFE', 'FL_KNEE'))


def test_joint_limits_come_from_the_urdf():
  """<limit lower upper> of the revolute joints -> SoloModel.joint_lower / joint_upper (the fixture's
  -10 / +10 rad, gym_solo/core/test_obs_observations.py:123-162 columns 8-9, when the file says so)."""
  text = to_urdf()
  m = parse_urdf(text).to_abi()
  assert list(m.joint_lower) == [-10.0] * 8 and list(m.joint_upper) == [10.0] * 8
  tight = text.replace('<limit lower="-10" upper="10" effort="1000" velocity="1000"/>',
                       '<limit lower="-1.5" upper="2.5" effort="1000" velocity="1000"/>', 1)
  m = parse_urdf(tight).to_abi()
  assert (m.joint_lower[0], m.joint_upper[0]) == (-1.5, 2.5) and m.joint_lower[1] == -10.0
  with pytest.raises(ValueError):
    parse_urdf(text.replace('lower="-10" upper="10"', 'lower="3" upper="-3"', 1)).to_abi()


def odri_style_urdf():
  """A Solo8 description written the way the Open Dynamic Robot Initiative's xacro output looks - NOT the loader's own
  to_urdf(): <visual> and <collision> MESHES with package:// paths on every link (two collisions on the base), material
  and <dynamics> elements, the HFE joints `continuous` (no position limits in the file), the FL upper leg's <inertial>
  given in a frame rotated by 90 degrees about z (the tensor written in that frame) - with the inertial numbers of the
  build's model, so that the reference's getJointInfo fixture must still come out."""
  from gym_solo_amd.model import LEGS
  m = Solo8Model()
  def inertial(li, rpy=(0.0, 0.0, 0.0)):
    I = np.asarray(li.inertia)
    if rpy[2] != 0.0:   # the file's tensor is expressed in the rotated frame: I_file = R^T I R
      c, s_ = np.cos(rpy[2]), np.sin(rpy[2])
      Rz = np.array([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]])
      I = Rz.T @ I @ Rz
    return ('    <inertial>\n      <origin xyz="%r %r %r" rpy="%r %r %r"/>\n      <mass value="%r"/>\n'
            '      <inertia ixx="%r" ixy="%r" ixz="%r" iyy="%r" iyz="%r" izz="%r"/>\n    </inertial>\n' % (
              *[float(x) for x in li.com], *[float(x) for x in rpy], float(li.mass),
              float(I[0, 0]), float(I[0, 1]), float(I[0, 2]), float(I[1, 1]), float(I[1, 2]), float(I[2, 2])))
  def shapes(mesh, n_collisions=1):
    vis = ('    <visual>\n      <origin xyz="0 0 0" rpy="0 0 0"/>\n      <geometry><mesh filename="package://robot_properties_solo/meshes/stl/%s"/></geometry>\n'
           '      <material name="grey"><color rgba="0.8 0.8 0.8 1.0"/></material>\n    </visual>\n' % mesh)
    col = ('    <collision>\n      <origin xyz="0 0 0" rpy="0 0 0"/>\n      <geometry><mesh filename="package://robot_properties_solo/meshes/stl/%s"/></geometry>\n'
           '    </collision>\n' % mesh)
    return vis + col * n_collisions
  out = ['<?xml version="1.0" ?>', '<robot name="solo" xmlns:xacro="http://www.ros.org/wiki/xacro">',
         '  <link name="base_link">\n' + inertial(m.base()) + shapes('solo_body.stl', 2) + '  </link>']
  for leg, L in enumerate(LEGS):
    side = 'left' if L[1] == 'L' else 'right'
    out.append('  <joint name="%s_HFE" type="continuous">\n    <parent link="base_link"/>\n    <child link="%s_UPPER_LEG"/>\n'
               '    <limit effort="1000" velocity="1000"/>\n    <axis xyz="0 1 0"/>\n    <origin xyz="%r %r %r" rpy="0 0 0"/>\n'
               '    <dynamics damping="0.0" friction="0.0"/>\n  </joint>' % ((L, L) + tuple(float(x) for x in m.hip_origin(leg))))
    out.append('  <link name="%s_UPPER_LEG">\n' % L + inertial(m.upper(leg), (0.0, 0.0, np.pi / 2) if L == 'FL' else (0.0, 0.0, 0.0)) +
               shapes('with_foot/solo_upper_leg_%s_side.stl' % side) + '  </link>')
    out.append('  <joint name="%s_KFE" type="revolute">\n    <parent link="%s_UPPER_LEG"/>\n    <child link="%s_LOWER_LEG"/>\n'
               '    <limit effort="1000" lower="-10" upper="10" velocity="1000"/>\n    <axis xyz="0 1 0"/>\n    <origin xyz="%r %r %r" rpy="0 0 0"/>\n'
               '    <dynamics damping="0.0" friction="0.0"/>\n  </joint>' % ((L, L, L) + tuple(float(x) for x in m.knee_origin(leg))))
    out.append('  <link name="%s_LOWER_LEG">\n' % L + inertial(m.lower(leg)) + shapes('with_foot/solo_lower_leg_%s_side.stl' % side) + '  </link>')
    out.append('  <joint name="%s_ANKLE" type="fixed">\n    <parent link="%s_LOWER_LEG"/>\n    <child link="%s_FOOT"/>\n'
               '    <origin xyz="%r %r %r" rpy="0 0 0"/>\n  </joint>' % ((L, L, L) + tuple(float(x) for x in m.ankle_origin(leg))))
    out.append('  <link name="%s_FOOT">\n' % L + inertial(m.foot(leg)) + shapes('with_foot/solo_foot.stl') + '  </link>')
  return '\n'.join(out + ['</robot>', ''])


def test_mesh_collisions_are_reported_not_silently_replaced():
  """Round 4's loader fell back to the built-in spheres without a word whenever a <collision> was not a sphere, and had
  only ever parsed its own to_urdf() output.  An ODRI-style file (mesh collisions with package:// paths, a rotated
  inertial frame, continuous hip joints): the model comes out - the reference's getJointInfo fixture included -, ONE
  warning names every link whose geometry was ignored, and the model says which spheres and limits are assumptions."""
  import warnings
  text = odri_style_urdf()
  with warnings.catch_warnings(record=True) as caught:
    warnings.simplefilter('always')
    model = parse_urdf(text)
  geo = [w for w in caught if issubclass(w.category, UrdfGeometryWarning)]
  assert len(geo) == 1
  msg = str(geo[0].message)
  for link in ['base_link'] + [L + suffix for L in ('FL', 'FR', 'HL', 'HR') for suffix in ('_UPPER_LEG', '_LOWER_LEG', '_FOOT')]:
    assert link in msg, link
  assert 'package://robot_properties_solo/meshes/stl/solo_body.stl' in msg and '16 built-in sphere(s)' in msg
  assert len(model.ignored_collisions) == 13 and len(model.ignored_collisions['base_link']) == 2
  assert model.sphere_sources == ['built-in'] * 16
  assert model.assumed_limits == ['FL_HFE', 'FR_HFE', 'HL_HFE', 'HR_HFE']
  # the model: identical to the built-in one (the rotated inertial frame undone, the continuous joints at +-10 rad)
  a, b = Solo8Model().to_abi(), model.to_abi()
  for field, *_ in abi.SoloModel._fields_:
    np.testing.assert_allclose(np.ctypeslib.as_array(getattr(b, field)) if hasattr(getattr(b, field), '_length_') else getattr(b, field),
                               np.ctypeslib.as_array(getattr(a, field)) if hasattr(getattr(a, field), '_length_') else getattr(a, field),
                               rtol=0, atol=1e-15, err_msg=field)
  # ... and the reference's fixture (gym_solo/core/test_obs_observations.py:123-162) is still reproduced from this file
  fixture = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'joint_info_fixture.json')))
  for f, o in zip(fixture['joint_info'], pybullet_joint_info(model)):
    assert o[1].decode() == f[1] and o[2] == f[2] and o[16] == f[16]
    np.testing.assert_allclose(o[13], f[13], atol=2e-5)
    np.testing.assert_allclose(o[14], f[14], atol=2e-6)
    np.testing.assert_allclose(o[15], f[15], atol=2e-5)
  # the loader's own output (sphere collisions on the feet) raises no warning and takes the four foot spheres from the file
  with warnings.catch_warnings():
    warnings.simplefilter('error', UrdfGeometryWarning)
    own = parse_urdf(to_urdf())
  assert own.sphere_sources.count('urdf') == 4 and not own.ignored_collisions and own.assumed_limits == []
