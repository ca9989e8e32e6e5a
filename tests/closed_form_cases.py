"""Closed forms of the CONSTRAINT rows - Coulomb friction, the saturated POSITION_CONTROL motor, the penetration
push-out - that hold whatever the solver's inner workings are ("pin the constraint rows with closed forms that do
not go through the oracle's step").  Each case is a set of initial states plus a checker of the state(s) a `stepSimulation`
(gym_solo/envs/solo8v2vanilla.py:87-91) leaves; tests/test_oracle_physics.py holds the CPU oracle to them (a few robots),
tests/test_gpu_closed_forms.py the HIP engine (4096 robots, through the C-ABI).  The checkers use Newton's law, Coulomb's law
and the model's masses / geometry only; of the oracle they use the KINEMATIC helpers (`momentum`, and `step_debug`'s CRBA
mass matrix M(q)) as measuring devices, never a step's result.

Conventions: state records [N, 32] (include/solo_engine.h), world-frame base velocities, generalised velocity
u = [R^T w, R^T v, qd] (base-body coordinates)."""
import numpy as np

from gym_solo_amd import abi
from gym_solo_amd.model import Solo8Model

THETA = np.radians(10.0)                                    # helpers.incline_terrain(10): h = tan(theta) x, uphill = +x
N_SLOPE = np.array([-np.sin(THETA), 0.0, np.cos(THETA)])    # ground normal
T1_SLOPE = np.array([np.cos(THETA), 0.0, np.sin(THETA)])    # world x projected into the tangent plane: UP the slope
VEL = (slice(abi.S_ANGVEL, abi.S_ANGVEL + 3), slice(abi.S_LINVEL, abi.S_LINVEL + 3), slice(abi.S_QD, abi.S_QD + 8))


def quat_about_y(angle):
  return np.array([0.0, np.sin(angle / 2), 0.0, np.cos(angle / 2)])


def rot(q):
  x, y, z, w = q
  return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                   [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                   [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def with_velocities(pre, post):
  """the configuration of `pre` with the velocities of `post`: the momentum of THAT state minus the momentum of `pre` is
  the impulse the step applied (a step changes velocities at the configuration it started from)"""
  s = pre.copy()
  for sl in VEL:
    s[sl] = post[sl]
  return s


# ---- (a) Coulomb on the incline ------------------------------------------------------------------------------------------
def standing_on_incline(n, lift=1e-3):
  """n robots standing (all joints 0: the motors hold the pose against targets 0) on the 10-degree incline, the base pitched
  with the slope so that the four feet meet the ground together; nothing but the feet is near the ground."""
  model = Solo8Model()
  st = np.zeros((n, abi.STATE_STRIDE))
  st[:, abi.S_POS:abi.S_POS + 3] = N_SLOPE * (0.32 + model.foot_radius + lift)
  st[:, abi.S_QUAT:abi.S_QUAT + 4] = quat_about_y(-THETA)
  return st


def incline_frictions(n, seed=0):
  """per-robot friction: the first half slides (mu < tan 10 deg = 0.176), the second half sticks"""
  rng = np.random.default_rng(seed)
  return np.concatenate([rng.uniform(0.02, 0.15, n // 2), rng.uniform(0.25, 1.0, n - n // 2)])


def check_coulomb_step(momentum, pre, post, mu, dt, g=9.81):
  """One step of one robot on the incline, every foot sliding DOWN the slope (friction rows along t1 saturated at + mu x
  their normal impulse): contact i applies lam_i (n + mu t1) + (its t2 impulse) t2, so the component of the step's total
  external impulse along w = t1 - mu n - perpendicular to all of that - is gravity's alone:
      (p_after - p_before) . w = - m g dt (sin theta - mu cos theta)
  however the normal load is shared between the feet.  Returns (measured, expected)."""
  m = Solo8Model().total_mass
  dp = momentum(with_velocities(pre, post)) - momentum(pre)
  w = T1_SLOPE - mu * N_SLOPE
  return float(dp @ w), -m * g * dt * (np.sin(THETA) - mu * np.cos(THETA))


# ---- (b) the saturated motor row -----------------------------------------------------------------------------------------
def floating_at_rest(n, seed=1):
  """n robots afloat (no contact) AT REST in random poses - no velocity, so no velocity-product forces: with zero gravity
  the only thing that acts in a step is the joint motors - and motor targets [n, 12]: per dof either FAR away (+-3 ... 6 rad:
  the row's target velocity kp (q* - q) / dt is hundreds of rad/s, far beyond what +-limit dt of impulse can give: saturated)
  or the joint's own angle (target velocity 0: the row holds the joint still with whatever impulse that takes)."""
  rng = np.random.default_rng(seed)
  st = np.zeros((n, abi.STATE_STRIDE))
  st[:, abi.S_POS + 2] = 2.0
  q = rng.normal(size=(n, 4))
  st[:, abi.S_QUAT:abi.S_QUAT + 4] = q / np.linalg.norm(q, axis=1, keepdims=True)
  st[:, abi.S_Q:abi.S_Q + 8] = rng.uniform(-1.5, 1.5, (n, 8))
  far = rng.random((n, 8)) < 0.6
  sign = np.where(rng.random((n, 8)) < 0.5, -1.0, 1.0)
  tgt = st[:, abi.S_Q:abi.S_Q + 8] + np.where(far, sign * rng.uniform(3.0, 6.0, (n, 8)), 0.0)
  acts = np.zeros((n, abi.NUM_JOINTS))
  for d in range(abi.NUM_DOF):
    acts[:, 3 * (d // 2) + d % 2] = tgt[:, d]
  return st, acts, far, sign


def check_motor_clamp(M, pre, post, far, sign, limit_dt):
  """M(q) du = J^T lam: with motor rows only, the generalised impulse of a step is 0 on the six base rows and lam_j on joint j.
  A row with a far target is SATURATED: lam_j = + limit dt towards the target.  A row whose target is the joint's own angle
  wants the joint at rest after the step (target velocity 0, kd = 1) and gets there unless the bound stops it
  (complementarity of a box-constrained row): either qd_j = 0 with |lam_j| <= limit dt, or lam_j = - sign(qd_j) limit dt.
  M: the oracle's CRBA mass matrix at `pre` (kinematics, not a step).  Returns the worst violations
  (base rows, saturated rows, holding rows) in impulse units, and how many holding rows held / were at their bound."""
  R = rot(pre[abi.S_QUAT:abi.S_QUAT + 4])
  def gen(s):
    return np.concatenate([R.T @ s[VEL[0]], R.T @ s[VEL[1]], s[VEL[2]]])
  imp = M @ (gen(post) - gen(pre))
  lam, qd = imp[6:], post[VEL[2]]
  held = ~far & (np.abs(qd) < 1e-9)
  stopped = ~far & ~held
  worst_hold = 0.0
  if held.any():
    worst_hold = max(worst_hold, float(np.maximum(np.abs(lam[held]) - limit_dt, 0.0).max()))
  if stopped.any():
    worst_hold = max(worst_hold, float(np.abs(lam[stopped] + np.sign(qd[stopped]) * limit_dt).max()))
  return (float(np.abs(imp[:6]).max()), float(np.abs(lam[far] - sign[far] * limit_dt).max()) if far.any() else 0.0, worst_hold,
          int(held.sum()), int(stopped.sum()))


# ---- (c) the penetration push-out (erp) ----------------------------------------------------------------------------------
def belly_corner_penetrating(n, seed=2):
  """n robots AT REST, legs pointing up (nothing of a leg near the ground), the base rolled and pitched (0.12 ... 0.3 rad each,
  either sign) so that ONE bottom corner sphere of the base is the lowest point, penetrating the flat ground by d (0.05 ... 2 mm,
  per robot).  Returns (states, actions, depths, per robot: that sphere's centre in the base frame, the radius)."""
  model = Solo8Model()
  rng = np.random.default_rng(seed)
  base = [(c, r) for b, c, r in model.spheres() if b == 0]
  st = np.zeros((n, abi.STATE_STRIDE))
  d = rng.uniform(5e-5, 2e-3, n)
  centres = np.zeros((n, 3))
  for e in range(n):
    roll, pitch = rng.uniform(0.12, 0.3, 2) * np.where(rng.random(2) < 0.5, -1.0, 1.0)
    cr, sr, cp, sp = np.cos(roll / 2), np.sin(roll / 2), np.cos(pitch / 2), np.sin(pitch / 2)
    q = np.array([sr * cp, cr * sp, -sr * sp, cr * cp])              # pybullet's getQuaternionFromEuler(roll, pitch, 0)
    zs = [(rot(q) @ c)[2] - r for c, r in base]
    low = int(np.argmin(zs))
    centres[e] = base[low][0]
    st[e, abi.S_QUAT:abi.S_QUAT + 4] = q
    st[e, abi.S_POS + 2] = -d[e] - zs[low]
  st[:, abi.S_Q:abi.S_Q + 8] = np.tile([np.pi, 0.0], 4)              # legs straight up
  acts = np.zeros((n, abi.NUM_JOINTS))
  for leg in range(4):
    acts[:, 3 * leg] = np.pi
  return st, acts, d, centres, model.base_sphere_radius


def contact_point_velocity(post, pre, centre, radius):
  """world velocity, after the step, of the body-fixed point that was the sphere's contact point (centre - r z) at `pre`"""
  R = rot(pre[abi.S_QUAT:abi.S_QUAT + 4])
  arm = R @ centre - radius * np.array([0.0, 0.0, 1.0])
  return post[VEL[1]] + np.cross(post[VEL[0]], arm)


# ---- (d) the base link keeps its own friction ------------------------------------------------------------------------------
def belly_on_incline(n, lift=5e-4):
  """n robots AT REST lying on the four bottom corner spheres of the base (legs straight up) on the 10-degree incline, the
  base pitched with the slope: the only contacts are the BASE link's.  gym_solo sets lateralFriction for links 0 .. 11 only
  (solo8v2vanilla.py:157-163): with lateral_friction = 0.1 < tan 10 deg the belly still has 0.5 and the robot stays."""
  model = Solo8Model()
  hz, r = model.base_sphere_half_extents[2], model.base_sphere_radius
  st = np.zeros((n, abi.STATE_STRIDE))
  st[:, abi.S_POS:abi.S_POS + 3] = N_SLOPE * (hz + r + lift)
  st[:, abi.S_QUAT:abi.S_QUAT + 4] = quat_about_y(-THETA)
  st[:, abi.S_Q:abi.S_Q + 8] = np.tile([np.pi, 0.0], 4)
  acts = np.zeros((n, abi.NUM_JOINTS))
  for leg in range(4):
    acts[:, 3 * leg] = np.pi
  return st, acts


# ---- (e) the joint-limit row ([recalled] btMultiBodyJointLimitConstraint; limits: the reference's getJointInfo fixture) -------
def joints_running_into_limits(n, limit=10.0, margin=0.5, seed=4):
  """n robots afloat, no gravity, motors WITHOUT torque (motor_torque_limit = 0: their rows are pinned to zero impulse), at rest
  but for ONE joint per robot, which sits C inside a limit (0.01 ... 0.3 rad, within joint_limit_margin) and moves towards it at s
  rad/s - some slow enough to stay inside within the step (s dt < C: the row must do NOTHING), some fast enough to cross it (the
  row must stop them at the speed that just reaches the limit, C / dt).  Returns (states, dof, side (+1 upper / -1 lower), C, s)."""
  rng = np.random.default_rng(seed)
  st = np.zeros((n, abi.STATE_STRIDE))
  st[:, abi.S_POS + 2] = 2.0
  q = rng.normal(size=(n, 4))
  st[:, abi.S_QUAT:abi.S_QUAT + 4] = q / np.linalg.norm(q, axis=1, keepdims=True)
  st[:, abi.S_Q:abi.S_Q + 8] = rng.uniform(-1.5, 1.5, (n, 8))
  dof = rng.integers(0, 8, n)
  side = np.where(rng.random(n) < 0.5, -1.0, 1.0)
  c = rng.uniform(0.01, 0.3, n)
  fast = rng.random(n) < 0.6
  s = np.where(fast, rng.uniform(1.2, 4.0, n), rng.uniform(0.05, 0.9, n)) * c / 1e-3   # relative to C / dt (dt = 1e-3)
  st[np.arange(n), abi.S_Q + dof] = side * (limit - c)
  st[np.arange(n), abi.S_QD + dof] = side * s
  return st, dof, side, c, s


def check_joint_limit_against_free(M, pre, post, free, dof, side, c, dt):
  """The joint's rate TOWARDS its limit after the step is min(what it would be without the row, C / dt) - a unilateral row in
  speculative form: it acts only if the joint would cross the limit within the step, and then leaves it exactly ON the limit - and
  what the row adds to the step's generalised impulse, M(q) (u_post - u_free), is zero everywhere but on that joint's row, where it
  pushes AWAY from the limit.  `free` = the state after the same step from the same state with the limits far away (a single moving
  joint has velocity-product forces: they are in both).  M: the oracle's CRBA mass matrix at `pre`.
  Returns (rate error, largest off-row impulse, sign violation, whether the row acted)."""
  R = rot(pre[abi.S_QUAT:abi.S_QUAT + 4])
  def gen(s):
    return np.concatenate([R.T @ s[VEL[0]], R.T @ s[VEL[1]], s[VEL[2]]])
  imp = M @ (gen(post) - gen(free))             # what the limit row added to the step
  towards_free = side * free[abi.S_QD + dof]    # the rate towards the limit without the row
  towards = side * post[abi.S_QD + dof]
  want = min(towards_free, c / dt)
  lam = -side * imp[6 + dof]                    # the row's impulse, counted AWAY from the limit
  off = np.abs(np.delete(imp, 6 + dof)).max()
  return abs(towards - want), off, max(0.0, -lam), (lam > 1e-12)


# ---- (f) link damping ([recalled] btMultiBody: - m v k (1 + |v|) per link; configs.py:21-22 set k for every link) ------------------
def translating_afloat(n, seed=6):
  """n robots afloat in random poses, no gravity, no joint motion, no rotation: every link moves with the same velocity v0, so the
  total damping force is - m_total k (1 + |v0|) v0 and one explicit-Euler step leaves v0 (1 - dt k (1 + |v0|)), nothing else."""
  rng = np.random.default_rng(seed)
  st = np.zeros((n, abi.STATE_STRIDE))
  st[:, abi.S_POS + 2] = 2.0
  q = rng.normal(size=(n, 4))
  st[:, abi.S_QUAT:abi.S_QUAT + 4] = q / np.linalg.norm(q, axis=1, keepdims=True)
  st[:, abi.S_Q:abi.S_Q + 8] = rng.uniform(-1.5, 1.5, (n, 8))
  st[:, abi.S_LINVEL:abi.S_LINVEL + 3] = rng.uniform(-3.0, 3.0, (n, 3))
  acts = np.zeros((n, abi.NUM_JOINTS))
  for d in range(abi.NUM_DOF):
    acts[:, 3 * (d // 2) + d % 2] = st[:, abi.S_Q + d]     # the motors hold the pose (nothing for them to do)
  return st, acts
