"""Solo8 (solo8v2) rigid-body model constants.

The reference loads ``assets/solo8v2/solo.urdf`` (gym_solo/envs/solo8v2vanilla.py:20,151-155)
from a git submodule that is EMPTY in the reference checkout (.gitmodules:1-3), so the model
is rebuilt here from

* what the reference's own test fixture pins — the 12-row ``getJointInfo`` dump in
  gym_solo/core/test_obs_observations.py:123-162 (tree topology, joint axes, joint frame
  positions/orientations in principal-axis inertial frames, limits), and
* [recalled] values of the public ODRI Solo8 URDF for what the fixture cannot pin (link
  masses, absolute inertias, base inertia, foot) — ASSUMPTIONS, flagged below, and
* collision geometry chosen by this build (spheres; the reference's meshes are absent) —
  ASSUMPTIONS.

``pybullet_joint_info()`` re-derives the fixture rows from these constants (principal-axis
frames, as ``URDF_USE_INERTIA_FROM_FILE`` produces) so tests/test_model.py can check every
fixture digit the constants are responsible for.
"""
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

from gym_solo_amd import abi

LEGS = ('FL', 'FR', 'HL', 'HR')
JOINT_NAMES = [f'{leg}_{j}' for leg in LEGS for j in ('HFE', 'KFE', 'ANKLE')]
LINK_NAMES = [f'{leg}_{j}' for leg in LEGS for j in ('UPPER_LEG', 'LOWER_LEG', 'FOOT')]
# pybullet joint type ids seen in the fixture: 0 = revolute, 4 = fixed
JOINT_TYPES = [0, 0, 4] * 4
# dof j (FL_HFE, FL_KFE, FR_HFE, ...) -> pybullet joint index
DOF_TO_JOINT = [3 * (j // 2) + (j % 2) for j in range(abi.NUM_DOF)]
JOINT_TO_DOF = {jt: d for d, jt in enumerate(DOF_TO_JOINT)}
JOINT_LIMIT = 10.0        # fixture cols 8, 9
JOINT_MAX_FORCE = 1000.0  # fixture col 10
JOINT_MAX_VEL = 1000.0    # fixture col 11


def _sym(xx, yy, zz, xy=0.0, xz=0.0, yz=0.0):
  return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]], dtype=np.float64)


@dataclass
class LinkInertial:
  mass: float
  com: np.ndarray       # in link (joint) frame
  inertia: np.ndarray   # 3x3 about com, link axes


def _merge(a: LinkInertial, b: LinkInertial) -> LinkInertial:
  """Rigidly weld two bodies expressed in the same frame."""
  m = a.mass + b.mass
  c = (a.mass * a.com + b.mass * b.com) / m
  def shift(I, mass, d):
    return I + mass * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
  I = shift(a.inertia, a.mass, a.com - c) + shift(b.inertia, b.mass, b.com - c)
  return LinkInertial(m, c, I)


@dataclass
class Solo8Model:
  """URDF-level description (link frames = joint frames, all joint axes +y)."""
  # fixture-pinned: HFE joint origins in the base frame (test_obs_observations.py:125,137,...)
  hip_x: float = 0.19
  hip_y: float = 0.1046
  # [recalled] ODRI URDF, cross-checked against the fixture's KFE/ANKLE parentFramePos
  knee_offset: Tuple[float, float, float] = (0.0, 0.03745, -0.16)
  ankle_offset: Tuple[float, float, float] = (0.0, 0.008, -0.16)
  upper_com: Tuple[float, float, float] = (1.377e-05, 0.01935853, -0.078707)
  lower_com: Tuple[float, float, float] = (0.0, 0.00787644, -0.08928215)
  # the fixture pins the principal-axis tilt, i.e. Iyz given (Iyy - Izz)
  upper_I: Tuple[float, float, float, float] = (4.1107e-4, 4.1193e-4, 3.024e-5, 4.671e-05)
  lower_I: Tuple[float, float, float, float] = (1.2024e-4, 1.2029e-4, 2.16e-6, 3.05e-06)
  # ASSUMPTIONS ([recalled], not pinned by anything in the reference)
  base_mass: float = 1.16115091
  base_I: Tuple[float, float, float] = (0.00578574, 0.01938108, 0.02476124)
  upper_mass: float = 0.14853845
  lower_mass: float = 0.03070001
  foot_mass: float = 0.00693606
  foot_com: Tuple[float, float, float] = (0.0, 0.0, 0.0)
  foot_I: float = 8.0e-7
  # ASSUMPTIONS: collision spheres (meshes absent). foot radius chosen so that the standing
  # height at q = 0 is 0.32 + r = 0.33698, the target height used by
  # examples/solo8_vanilla/interactive_pos_control.py:23
  foot_radius: float = 0.01698
  # CALIBRATED on the one pybullet-extracted state the reference holds - the 12 getJointState rows
  # "at rest" of gym_solo/core/test_obs_observations.py:256-275: |HFE| = 1.53013, |KFE| = 3.08532,
  # joint rates ~1e-11.  At 2 N.m four saturated hip motors would press the knees down with
  # 4 x 2 / 0.16 = 50 N and lift the 1.9 kg robot, so a state at rest 0.041 / 0.056 rad short of the
  # folded targets is a PASSIVE rest: belly on the ground, every leg lying on its knee and its foot,
  # the motors not carrying the links (an older, weaker-motor configuration: the vector's sign
  # pattern predates today's starting_joint_pos as well).  Then the angles are pure geometry:
  #   hip height - knee contact radius = 0.16 sin(pi/2 - 1.53013)              = 6.50 mm
  #   hip height - foot contact radius = 6.50 mm + 0.16 sin(3.08532 - 1.53013 - pi/2) = 9.00 mm
  # With the foot radius above: belly-to-hip height 0.02598 m and knee radius 0.01948 m
  # (tests/test_oracle_physics.py::test_passive_rest_pose_reproduces_the_reference_vector, also on
  # the HIP engine).  A calibration of two free collision parameters, not a validation.
  knee_radius: float = 0.01948
  base_sphere_radius: float = 0.02
  base_sphere_half_extents: Tuple[float, float, float] = (0.19, 0.055, 0.00598)

  # ---- per-leg mirrored quantities -------------------------------------------------
  def leg_signs(self, leg: int) -> Tuple[float, float]:
    sx = 1.0 if leg < 2 else -1.0      # F / H
    sy = 1.0 if leg % 2 == 0 else -1.0  # L / R
    return sx, sy

  def hip_origin(self, leg: int) -> np.ndarray:
    sx, sy = self.leg_signs(leg)
    return np.array([sx * self.hip_x, sy * self.hip_y, 0.0])

  def knee_origin(self, leg: int) -> np.ndarray:
    _, sy = self.leg_signs(leg)
    return np.array([self.knee_offset[0], sy * self.knee_offset[1], self.knee_offset[2]])

  def ankle_origin(self, leg: int) -> np.ndarray:
    _, sy = self.leg_signs(leg)
    return np.array([self.ankle_offset[0], sy * self.ankle_offset[1], self.ankle_offset[2]])

  def upper(self, leg: int) -> LinkInertial:
    _, sy = self.leg_signs(leg)
    c = np.array([sy * self.upper_com[0], sy * self.upper_com[1], self.upper_com[2]])
    xx, yy, zz, yz = self.upper_I
    return LinkInertial(self.upper_mass, c, _sym(xx, yy, zz, yz=sy * yz))

  def lower(self, leg: int) -> LinkInertial:
    _, sy = self.leg_signs(leg)
    # asset quirk pinned by the fixture (FR_ANKLE / HR_ANKLE parentFramePos y = -0.0140471,
    # test_obs_observations.py:141-143,159-161): the right lower-leg CoM y is NOT mirrored.
    c = np.array([self.lower_com[0], self.lower_com[1], self.lower_com[2]])
    xx, yy, zz, yz = self.lower_I
    return LinkInertial(self.lower_mass, c, _sym(xx, yy, zz, yz=sy * yz))

  def foot(self, leg: int) -> LinkInertial:
    return LinkInertial(self.foot_mass, np.array(self.foot_com, dtype=np.float64),
                        np.eye(3) * self.foot_I)

  def lower_with_foot(self, leg: int) -> LinkInertial:
    f = self.foot(leg)
    f_in_lower = LinkInertial(f.mass, f.com + self.ankle_origin(leg), f.inertia)
    return _merge(self.lower(leg), f_in_lower)

  @property
  def total_mass(self) -> float:
    return self.base_mass + 4 * (self.upper_mass + self.lower_mass + self.foot_mass)

  # ---- flattened for the C-ABI -----------------------------------------------------
  def base(self) -> LinkInertial:
    return LinkInertial(self.base_mass, np.zeros(3), np.diag(self.base_I))

  def spheres(self):
    """Collision spheres as (body, centre in body frame, radius), 4 per leg l in contact-solve
    order: 4l knee (lower-leg origin), 4l+1 foot (ankle origin), 4l+2 / 4l+3 the two base-box
    corners (bottom, top) next to that leg's hip.  body: 0 = base, 1+2l = upper, 2+2l = lower."""
    hx, hy, hz = self.base_sphere_half_extents
    out = []
    for leg in range(abi.NUM_LEGS):
      sx, sy = self.leg_signs(leg)
      out.append((2 + 2 * leg, np.zeros(3), self.knee_radius))
      out.append((2 + 2 * leg, self.ankle_origin(leg), self.foot_radius))
      for sz in (-1, 1):
        out.append((0, np.array([sx * hx, sy * hy, sz * hz]), self.base_sphere_radius))
    return out

  def joint_limits(self):
    """(lower, upper) [rad] per dof: the fixture's -10 / +10 (test_obs_observations.py:123-162)."""
    return [(-JOINT_LIMIT, JOINT_LIMIT)] * abi.NUM_DOF

  # ---- flattened for the C-ABI -----------------------------------------------------
  def to_abi(self) -> abi.SoloModel:
    return model_to_abi(self)


def model_to_abi(model) -> abi.SoloModel:
  """Flatten any object with the Solo8 accessor interface (base(), upper(leg), lower_with_foot(leg),
  hip_origin(leg), knee_origin(leg), spheres()) — Solo8Model or urdf.UrdfSolo8Model — into the
  C-ABI struct."""
  m = abi.SoloModel()
  def put6(dst, I):
    for k, (a, b) in enumerate(((0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2))):
      dst[k] = float(I[a, b])
  base = model.base()
  m.mass[0] = base.mass
  for a in range(3):
    m.com[0][a] = float(base.com[a])
  put6(m.inertia[0], base.inertia)
  for leg in range(abi.NUM_LEGS):
    ju, jl = 2 * leg, 2 * leg + 1
    bu, bl = 1 + ju, 1 + jl
    m.parent[ju] = 0
    m.parent[jl] = bu
    for a in range(3):
      m.joint_origin[ju][a] = model.hip_origin(leg)[a]
      m.joint_origin[jl][a] = model.knee_origin(leg)[a]
      m.joint_axis[ju][a] = (0.0, 1.0, 0.0)[a]
      m.joint_axis[jl][a] = (0.0, 1.0, 0.0)[a]
    for b, li in ((bu, model.upper(leg)), (bl, model.lower_with_foot(leg))):
      m.mass[b] = li.mass
      for a in range(3):
        m.com[b][a] = li.com[a]
      put6(m.inertia[b], li.inertia)
    m.dof_to_joint[ju] = DOF_TO_JOINT[ju]
    m.dof_to_joint[jl] = DOF_TO_JOINT[jl]
    limits = model.joint_limits() if hasattr(model, 'joint_limits') else [(-JOINT_LIMIT, JOINT_LIMIT)] * abi.NUM_DOF
    for j in (ju, jl):
      m.joint_lower[j], m.joint_upper[j] = float(limits[j][0]), float(limits[j][1])
  sph = model.spheres()
  if len(sph) != abi.MAX_SPHERES:
    raise ValueError('the engine expects {} collision spheres'.format(abi.MAX_SPHERES))
  for s, (body, center, radius) in enumerate(sph):
    m.sphere_body[s] = int(body)
    for a in range(3):
      m.sphere_center[s][a] = float(center[a])
    m.sphere_radius[s] = float(radius)
  m.num_spheres = len(sph)
  return m


def _principal_rotation(I: np.ndarray) -> np.ndarray:
  """Rotation Rp (link axes <- principal axes) closest to identity with Rp^T I Rp diagonal,
  i.e. the inertial frame pybullet builds under URDF_USE_INERTIA_FROM_FILE."""
  w, V = np.linalg.eigh(I)
  R = np.zeros((3, 3))
  used = set()
  for axis in range(3):
    k = max((j for j in range(3) if j not in used), key=lambda j: abs(V[axis, j]))
    used.add(k)
    col = V[:, k]
    R[:, axis] = col if col[axis] > 0 else -col
  if np.linalg.det(R) < 0:
    R[:, 2] = -R[:, 2]
  return R


def _quat_from_R(R: np.ndarray) -> np.ndarray:
  """xyzw quaternion of a rotation matrix (w > 0)."""
  w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
  x = (R[2, 1] - R[1, 2]) / (4 * w)
  y = (R[0, 2] - R[2, 0]) / (4 * w)
  z = (R[1, 0] - R[0, 1]) / (4 * w)
  return np.array([x, y, z, w])


def pybullet_joint_info(model: Solo8Model = None) -> List[tuple]:
  """The 12 tuples ``getJointInfo`` returns for this model (same 17 fields as the fixture,
  gym_solo/core/test_obs_observations.py:123-162). Frames are principal-axis inertial frames:
  axis = Rp_child^T a;  parentFramePos = Rp_parent^T (joint origin - parent com);
  parentFrameOrn = quat(Rp_child^T Rp_parent)  [field meaning recalled from the pybullet
  quickstart guide; verified against the fixture numerically]."""
  model = model or Solo8Model()
  limits = model.joint_limits() if hasattr(model, 'joint_limits') else [(-JOINT_LIMIT, JOINT_LIMIT)] * abi.NUM_DOF
  out = []
  for leg in range(abi.NUM_LEGS):
    up, lo, ft = model.upper(leg), model.lower(leg), model.foot(leg)
    Rb = np.eye(3)
    Ru, Rl, Rf = (_principal_rotation(x.inertia) for x in (up, lo, ft))
    chain = [
      (0, Rb, np.zeros(3), Ru, model.hip_origin(leg), -1),
      (0, Ru, up.com, Rl, model.knee_origin(leg), 3 * leg),
      (4, Rl, lo.com, Rf, model.ankle_origin(leg), 3 * leg + 1),
    ]
    for k, (jtype, Rp, pcom, Rc, origin, parent_idx) in enumerate(chain):
      idx = 3 * leg + k
      revolute = jtype == 0
      axis = tuple(Rc.T @ np.array([0.0, 1.0, 0.0])) if revolute else (0.0, 0.0, 0.0)
      pos = tuple(Rp.T @ (origin - pcom))
      orn = tuple(_quat_from_R(Rc.T @ Rp))
      dof = JOINT_TO_DOF.get(idx)
      q_index = 7 + dof if revolute else -1
      u_index = 6 + dof if revolute else -1
      lower, upper = limits[dof] if revolute else (-JOINT_LIMIT, JOINT_LIMIT)
      out.append((idx, JOINT_NAMES[idx].encode(), jtype, q_index, u_index, 1 if revolute else 0,
                  0.0, 0.0, float(lower), float(upper), JOINT_MAX_FORCE, JOINT_MAX_VEL,
                  LINK_NAMES[idx].encode(), axis, pos, orn, parent_idx))
  return out
