"""URDF <-> Solo8 model constants.

Counterpart of ``loadURDF(config.urdf, ..., flags=URDF_USE_INERTIA_FROM_FILE, useFixedBase=False)``
(gym_solo/envs/solo8v2vanilla.py:151-155, gym_solo/core/configs.py:36-38).  The reference's
``assets/solo8v2/solo.urdf`` lives in an empty git submodule, so users who have the real file can
load it here and get the engine's ``abi.SoloModel`` (and re-derive the reference's ``getJointInfo``
fixture from it with ``model.pybullet_joint_info``); ``to_urdf`` writes the built-in constants
out in the same format (used by the round-trip test).

Supported: the Solo8 topology — one base link and four legs ``{FL,FR,HL,HR}_{HFE,KFE}`` revolute
about +y plus a fixed ``*_ANKLE`` joint (welded into the lower leg); ``<inertial>`` with origin xyz/rpy;
``<collision>`` is only read when it is a ``<sphere>`` ON A FOOT LINK (meshes cannot be used by the
sphere/ground contact model; the built-in sphere set is used otherwise).

What the loader does NOT take from the file it SAYS (round 5): every ``<collision>`` that is not the one
sphere of a foot link - meshes (``package://`` paths are never opened), boxes, cylinders, second and
further collisions of a link, spheres on links whose sphere the model does not take from the file - is
recorded in ``UrdfSolo8Model.ignored_collisions`` and ``parse_urdf`` emits ONE ``UrdfGeometryWarning``
naming the links whose contact geometry is the built-in assumption of ``gym_solo_amd/model.py``; a
``continuous`` joint (no limits in the file) gets the built-in +-10 rad and is listed in
``assumed_limits``.
"""
import warnings
import xml.etree.ElementTree as ET

import numpy as np

from gym_solo_amd import abi
from gym_solo_amd.model import LEGS, LinkInertial, Solo8Model, _merge, _sym, model_to_abi


def _floats(text, n, default):
  if text is None:
    return np.array(default, dtype=np.float64)
  v = np.array([float(t) for t in text.split()], dtype=np.float64)
  if v.shape != (n,):
    raise ValueError('expected {} numbers, got {!r}'.format(n, text))
  return v


def _rpy_matrix(rpy):
  r, p, y = rpy
  cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
  return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                   [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                   [-sp, cp * sr, cp * cr]])


class UrdfGeometryWarning(UserWarning):
  """The URDF's collision geometry (or part of it) was not used: the built-in sphere set stands in for it."""


class UrdfSolo8Model:
  """Solo8 constants parsed from a URDF; same accessor interface as ``model.Solo8Model``.

  ``ignored_collisions``: {link name: [description of every <collision> the contact model did not take]};
  ``sphere_sources``: for each of the 16 collision spheres, 'urdf' or 'built-in';
  ``assumed_limits``: the revolute joints whose limits are the built-in +-10 rad (continuous, or no <limit lower upper>)."""

  def __init__(self, links, joints, fallback=None):
    self._links, self._joints = links, joints
    self._fallback = fallback or Solo8Model()
    children = {j['child'] for j in joints.values()}
    roots = [n for n in links if n not in children]
    if len(roots) != 1:
      raise ValueError('expected exactly one root (base) link, found {}'.format(roots))
    self.base_link = roots[0]
    for leg in LEGS:
      for jn, jtype in ((leg + '_HFE', 'revolute'), (leg + '_KFE', 'revolute'), (leg + '_ANKLE', 'fixed')):
        if jn not in joints:
          raise ValueError('missing joint {}'.format(jn))
        j = joints[jn]
        if j['type'] not in ((jtype, 'continuous') if jtype == 'revolute' else (jtype,)):
          raise ValueError('joint {} must be {}'.format(jn, jtype))
        if jtype == 'revolute' and not np.allclose(j['axis'], [0, 1, 0]):
          raise ValueError('joint {}: the engine is specialised to +y axes'.format(jn))
        if not np.allclose(j['rpy'], 0):
          raise ValueError('joint {}: rotated joint frames are not supported'.format(jn))
      if joints[leg + '_HFE']['parent'] != self.base_link:
        raise ValueError(leg + '_HFE must hang off the base link')
      if joints[leg + '_KFE']['parent'] != joints[leg + '_HFE']['child']:
        raise ValueError(leg + '_KFE must hang off the upper leg')
      if joints[leg + '_ANKLE']['parent'] != joints[leg + '_KFE']['child']:
        raise ValueError(leg + '_ANKLE must hang off the lower leg')
    # ---- what of the file's geometry is NOT used (the module's head says why)
    foot_links = {joints[leg + '_ANKLE']['child'] for leg in LEGS}
    self.ignored_collisions = {}
    for name, link in links.items():
      ignored = list(link.get('ignored', []))
      if link.get('sphere') is not None and name not in foot_links:
        ignored.insert(0, 'sphere (only a FOOT link\'s sphere is taken from the file)')
      if ignored:
        self.ignored_collisions[name] = ignored
    self.sphere_sources = ['built-in'] * abi.MAX_SPHERES
    for leg in range(abi.NUM_LEGS):
      if links[joints[LEGS[leg] + '_ANKLE']['child']].get('sphere') is not None:
        self.sphere_sources[4 * leg + 1] = 'urdf'
    self.assumed_limits = [jn for leg in LEGS for jn in (leg + '_HFE', leg + '_KFE')
                           if joints[jn]['type'] != 'revolute' or joints[jn].get('limit') is None]

  def geometry_report(self):
    """One paragraph: which links' collision geometry was ignored, and what stands in for it (None: nothing was)."""
    if not self.ignored_collisions:
      return None
    parts = ['{} ({})'.format(name, '; '.join(what)) for name, what in sorted(self.ignored_collisions.items())]
    n_file = sum(1 for s_ in self.sphere_sources if s_ == 'urdf')
    return ('URDF collision geometry NOT used by the sphere / ground contact model: ' + ', '.join(parts) +
            '.  Contact uses {} sphere(s) from the file and {} built-in sphere(s) of gym_solo_amd/model.py '
            '(an assumption, two of its parameters calibrated: DESIGN.md section 6) - mesh files are never opened.'.format(
              n_file, abi.MAX_SPHERES - n_file))

  def _inertial(self, link_name):
    return self._links[link_name]['inertial']

  def base(self):
    b = self._inertial(self.base_link)
    if not np.allclose(b.com, 0):
      raise ValueError('the base inertial origin must be the base link frame')
    return b

  def hip_origin(self, leg):
    return self._joints[LEGS[leg] + '_HFE']['xyz']

  def knee_origin(self, leg):
    return self._joints[LEGS[leg] + '_KFE']['xyz']

  def ankle_origin(self, leg):
    return self._joints[LEGS[leg] + '_ANKLE']['xyz']

  def upper(self, leg):
    return self._inertial(self._joints[LEGS[leg] + '_HFE']['child'])

  def lower(self, leg):
    return self._inertial(self._joints[LEGS[leg] + '_KFE']['child'])

  def foot(self, leg):
    return self._inertial(self._joints[LEGS[leg] + '_ANKLE']['child'])

  def lower_with_foot(self, leg):
    f = self.foot(leg)
    return _merge(self.lower(leg), LinkInertial(f.mass, f.com + self.ankle_origin(leg), f.inertia))

  @property
  def total_mass(self):
    return sum(l['inertial'].mass for l in self._links.values())

  def spheres(self):
    """Sphere collisions from the URDF where present (knee = on the lower leg at its origin, foot =
    on the foot link), the built-in assumptions otherwise."""
    out = list(self._fallback.spheres())
    for leg in range(abi.NUM_LEGS):
      foot_link = self._links[self._joints[LEGS[leg] + '_ANKLE']['child']]
      if foot_link.get('sphere') is not None:
        center, radius = foot_link['sphere']
        out[4 * leg + 1] = (2 + 2 * leg, self.ankle_origin(leg) + center, radius)
      else:
        out[4 * leg + 1] = (2 + 2 * leg, self.ankle_origin(leg), out[4 * leg + 1][2])
    return out

  def joint_limits(self):
    """<limit lower= upper=> of the eight revolute joints, dof order (continuous joints or missing
    limits: the built-in +-10 rad of the reference's fixture)."""
    out = list(self._fallback.joint_limits())
    for leg in range(abi.NUM_LEGS):
      for d, jn in enumerate((LEGS[leg] + '_HFE', LEGS[leg] + '_KFE')):
        lim = self._joints[jn].get('limit')
        if lim is not None and self._joints[jn]['type'] == 'revolute':
          if not lim[0] < lim[1]:
            raise ValueError('joint {}: limit lower must be below upper'.format(jn))
          out[2 * leg + d] = lim
    return out

  def to_abi(self):
    return model_to_abi(self)


def parse_urdf(text, fallback=None) -> UrdfSolo8Model:
  root = ET.fromstring(text)
  links, joints = {}, {}
  for el in root.findall('link'):
    ine = el.find('inertial')
    if ine is None:
      inertial = LinkInertial(0.0, np.zeros(3), np.zeros((3, 3)))
    else:
      org = ine.find('origin')
      xyz = _floats(org.get('xyz') if org is not None else None, 3, (0, 0, 0))
      rpy = _floats(org.get('rpy') if org is not None else None, 3, (0, 0, 0))
      i = ine.find('inertia')
      I = _sym(*(float(i.get(k, 0.0)) for k in ('ixx', 'iyy', 'izz', 'ixy', 'ixz', 'iyz')))
      Rm = _rpy_matrix(rpy)
      inertial = LinkInertial(float(ine.find('mass').get('value')), xyz, Rm @ I @ Rm.T)
    sphere, ignored = None, []
    for col in el.findall('collision'):
      geo = col.find('geometry')
      kind = geo[0].tag if geo is not None and len(geo) else 'empty'
      if kind == 'sphere' and sphere is None:
        corg = col.find('origin')
        sphere = (_floats(corg.get('xyz') if corg is not None else None, 3, (0, 0, 0)), float(geo[0].get('radius')))
      elif kind == 'mesh':
        ignored.append('mesh {}'.format(geo[0].get('filename', '?')))
      elif kind == 'sphere':
        ignored.append('a second sphere (one per link is read)')
      else:
        ignored.append(kind)
    links[el.get('name')] = {'inertial': inertial, 'sphere': sphere, 'ignored': ignored}
  for el in root.findall('joint'):
    org = el.find('origin')
    ax = el.find('axis')
    joints[el.get('name')] = {
      'type': el.get('type'), 'parent': el.find('parent').get('link'), 'child': el.find('child').get('link'),
      'xyz': _floats(org.get('xyz') if org is not None else None, 3, (0, 0, 0)),
      'rpy': _floats(org.get('rpy') if org is not None else None, 3, (0, 0, 0)),
      'axis': _floats(ax.get('xyz') if ax is not None else None, 3, (1, 0, 0)),
      'limit': (None if el.find('limit') is None or el.find('limit').get('lower') is None else
                (float(el.find('limit').get('lower')), float(el.find('limit').get('upper'))))}
  model = UrdfSolo8Model(links, joints, fallback)
  report = model.geometry_report()
  if report is not None:
    warnings.warn(report, UrdfGeometryWarning, stacklevel=2)
  return model


def load_urdf(path, fallback=None) -> UrdfSolo8Model:
  with open(path) as f:
    return parse_urdf(f.read(), fallback)


def to_urdf(model=None, name='solo') -> str:
  """Write a model with the accessor interface as URDF text (inertial data at 17 significant digits,
  sphere collisions for the feet)."""
  model = model or Solo8Model()
  def fmt(v):
    return ' '.join(repr(float(x)) for x in v)
  def link(lname, li, sphere=None):
    I = li.inertia
    s = ['  <link name="{}">'.format(lname), '    <inertial>',
         '      <origin xyz="{}" rpy="0 0 0"/>'.format(fmt(li.com)),
         '      <mass value="{!r}"/>'.format(float(li.mass)),
         '      <inertia ixx="{!r}" ixy="{!r}" ixz="{!r}" iyy="{!r}" iyz="{!r}" izz="{!r}"/>'.format(
           float(I[0, 0]), float(I[0, 1]), float(I[0, 2]), float(I[1, 1]), float(I[1, 2]), float(I[2, 2])),
         '    </inertial>']
    if sphere is not None:
      s += ['    <collision>', '      <origin xyz="{}" rpy="0 0 0"/>'.format(fmt(sphere[0])),
            '      <geometry><sphere radius="{!r}"/></geometry>'.format(float(sphere[1])), '    </collision>']
    return s + ['  </link>']
  def joint(jname, jtype, parent, child, xyz):
    s = ['  <joint name="{}" type="{}">'.format(jname, jtype),
         '    <parent link="{}"/>'.format(parent), '    <child link="{}"/>'.format(child),
         '    <origin xyz="{}" rpy="0 0 0"/>'.format(fmt(xyz))]
    if jtype == 'revolute':
      s += ['    <axis xyz="0 1 0"/>', '    <limit lower="-10" upper="10" effort="1000" velocity="1000"/>']
    return s + ['  </joint>']
  out = ['<?xml version="1.0"?>', '<robot name="{}">'.format(name)] + link('base_link', model.base())
  sph = model.spheres()
  for leg, L in enumerate(LEGS):
    foot_center = np.asarray(sph[4 * leg + 1][1]) - model.ankle_origin(leg)
    out += link(L + '_UPPER_LEG', model.upper(leg)) + link(L + '_LOWER_LEG', model.lower(leg))
    out += link(L + '_FOOT', model.foot(leg), sphere=(foot_center, sph[4 * leg + 1][2]))
    out += joint(L + '_HFE', 'revolute', 'base_link', L + '_UPPER_LEG', model.hip_origin(leg))
    out += joint(L + '_KFE', 'revolute', L + '_UPPER_LEG', L + '_LOWER_LEG', model.knee_origin(leg))
    out += joint(L + '_ANKLE', 'fixed', L + '_LOWER_LEG', L + '_FOOT', model.ankle_origin(leg))
  return '\n'.join(out + ['</robot>', ''])
