"""Environment configuration — batched counterpart of gym_solo/core/configs.py:8-38.

Every field of the reference's ``Solo8BaseConfig`` is kept with the same name, default and
"dataclass field vs plain class attribute" status (tests mutate ``max_motor_rotation`` on the
instance, gym_solo/envs/test_solo8v2vanilla.py:110).  Fields below the marker are build
extensions consumed by the HIP engine.
"""
from dataclasses import dataclass
from typing import Tuple

import numpy as np

from gym_solo_amd import abi


DEFAULT_ULP_TOLERANCE_F64, DEFAULT_ULP_TOLERANCE_F32 = 512, 2   # (Solo8BaseConfig.solver_ulp_tolerance = None)


@dataclass
class Solo8BaseConfig:
  dt: float = 1e-3
  # Max torque supplied by the motors
  motor_torque_limit: float = 2

  robot_start_pos: Tuple[float] = (0., 0., 0.5)
  robot_start_orientation_euler: Tuple[float] = (0., 0., 0.)
  gravity: Tuple[float] = (0., 0., -9.81)

  max_motor_rotation = 2 * np.pi

  linear_damping: float = .04
  angular_damping: float = .04
  restitution: float = 0.
  lateral_friction: float = 0.5

  # render_* fields of the reference (configs.py:26-34) configure the pybullet camera; rendering
  # is out of scope for the batched engine — kept so configs stay swappable.
  render_width: int = 369
  render_height: int = 369
  render_fov: int = 80
  render_aspect: float = render_width / render_height
  render_pos = [0, 0, .2]
  render_cam_distance = 1
  render_yaw = 0.
  render_pitch = -20.
  render_roll = 0.

  # ---- build extensions (not in the reference) -------------------------------------
  num_envs: int = 1
  device: int = 0
  # arithmetic type of the engine.  'float64' is what the reference computes in (PyBullet steps in double:
  # solo8v2vanilla.py:91) and the default of the drop-in; 'float32' is the documented OPT-IN fast mode (about 2x the
  # throughput; one-step joint-rate error ~1e-4 rad/s against the f64 path: DESIGN.md section 6)
  dtype: str = 'float64'
  solver_iterations: int = 50     # Bullet default [recalled]
  # Convergence of the Gauss-Seidel iteration: a row whose clamped candidate differs from its impulse by at most this many
  # half-ulps, relative (k x 2^-53 |impulse| in f64, k x 2^-24 in f32), is left alone, and a sweep that updates no row ends the
  # iteration (0 = only an exact fixed point ends it).  None = the precision's default: **512 in float64** - 5.7e-14 relative, half
  # of the rounding noise of the step itself (two f64 formulations of one step, the engine's and the oracle's, disagree by 1e-13 in
  # the median: tests/test_gpu_parity_scale.py measures it) - and 2 in float32 (1.2e-7).  Measured (round 6,
  # profiles/round6_ulp_tolerance_sweep.log): 0 / 2 / 512 / 4096 half-ulps leave a resting robot's joint rates and the 60-step
  # parity against the oracle's 50 plain sweeps where they are (5.5e-9 rad/s; 9e-11 / 2e-11 / 6e-11 - and 9e-10 at 4096), and
  # take the mean sweeps per robot-step from 9.7 to 9.5 / 8.4 / 8.0.  Round 5's default was 2.
  solver_ulp_tolerance: int = None
  # pybullet's solverResidualThreshold: the Gauss-Seidel iteration ends after a sweep whose largest squared
  # velocity-level change is below it.  pybullet's documented default is 1e-7 [recalled] and gym_solo never changes it -
  # but this solver does not warm-start, and with 1e-7 it leaves a resting robot jittering at 5e-5 rad/s, where the
  # reference's one pybullet-extracted state (test_obs_observations.py:256-275) rests at 1e-11: so OFF (0) by default,
  # the iteration runs to its fixed point; 1e-7 is the opt-in (what it is worth)
  solver_residual_threshold: float = 0.0
  # warm starting (an OPT-IN, only together with solver_residual_threshold > 0): a step's iteration starts from this
  # factor x the impulses the previous step ended with (clamped to this step's bounds) instead of from zero.  With it
  # the residual-threshold exit leaves a resting robot at rest (tests/test_gpu_warm_start.py: 1e-12 rad/s against
  # 5e-5 without).  [recalled] Bullet's rigid-body contacts use m_warmstartingFactor 0.85; 1.0 is the value that keeps
  # a fixed point fixed.  0 = off.
  solver_warm_start: float = 0.0
  motor_kp: float = 0.1           # pybullet POSITION_CONTROL default positionGain [recalled]
  motor_kd: float = 1.0           # pybullet POSITION_CONTROL default velocityGain [recalled]
  # friction of the BASE link's collision spheres.  gym_solo's load_bodies() calls changeDynamics(lateralFriction=...)
  # for `range(getNumJoints)` = links 0..11 (solo8v2vanilla.py:157-163) and never for the base (-1), which therefore keeps
  # pybullet's default 0.5 [recalled] whatever `lateral_friction` says; per-robot friction (Engine.set_params) leaves it alone too
  base_lateral_friction: float = 0.5
  contact_erp: float = 0.2
  contact_margin: float = 0.005
  joint_limit_margin: float = 0.5  # [rad] distance to a URDF joint limit below which its row is built
  settle_steps: int = 500         # gym_solo/envs/solo8v2vanilla.py:130
  auto_reset: bool = False
  # The launch geometry of rollouts.  -1 (the default of all three) = THE ENGINE CHOOSES, from what was measured on the
  # benchmark workload (solo_engine.hip: make_plan; Engine.plan(k) reports the choice): fused launches of min(K, 250)
  # steps; two batch slices on separate HIP streams when a rollout takes several launches; robot migration only when a
  # launch has more robots than the chip has wave slots (4096).  step() is always one launch of one step.
  steps_per_launch: int = -1      # rollouts fuse this many env steps per kernel launch (1: one launch per step)
  rollout_streams: int = -1       # rollouts advance this many batch slices on separate HIP streams
  # c > 0: a fused launch of more than c steps hands its robots from wave to wave every c steps through a work queue
  # in device memory, so that the launch ends when the work is done and not when the unluckiest SIMD's robots are
  # (scheduling only, results bit-identical).  0 = off: one wave steps one robot through the launch;
  # -1: the engine chooses (above)
  migrate_steps: int = -1
  # ground: None = pybullet_data's flat plane.urdf (solo8_base_env.py:47); or a heightfield
  # dict(heights=[ny, nx] array, cell=metres, origin=(x, y) of grid point (0, 0) or None = centred)
  terrain = None

  @property
  def urdf(self):
    """Path of the robot URDF if one is available, else None (configs.py:36-38 resolves a file
    packaged from the WPI-MMR/assets submodule, which is empty in the reference checkout; the
    engine then uses the built-in constants of gym_solo_amd/model.py).  ``urdf_path`` may be
    absolute or relative to the gym_solo_amd package."""
    import os
    path = getattr(self, 'urdf_path', None)
    if not path:
      return None
    for cand in (path, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path)):
      if os.path.isfile(cand):
        return cand
    return None


def euler_to_quat(euler) -> Tuple[float, float, float, float]:
  """pybullet.getQuaternionFromEuler: xyzw from roll, pitch, yaw (solo8v2vanilla.py:153)."""
  r, p, y = (float(e) for e in euler)
  cr, sr = np.cos(r / 2), np.sin(r / 2)
  cp, sp = np.cos(p / 2), np.sin(p / 2)
  cy, sy = np.cos(y / 2), np.sin(y / 2)
  return (sr * cp * cy - cr * sp * sy,
          cr * sp * cy + sr * cp * sy,
          cr * cp * sy - sr * sp * cy,
          cr * cp * cy + sr * sp * sy)


def config_to_abi(config, starting_joint_pos=None, joint_ordering=None,
                  normalize_actions=False) -> abi.SoloConfig:
  c = abi.SoloConfig()
  c.abi_version = abi.ABI_VERSION
  if config.dtype not in ('float32', 'float64'):
    raise ValueError('dtype must be float32 or float64: {}'.format(config.dtype))
  c.dtype = abi.F32 if config.dtype == 'float32' else abi.F64
  if not config.dt:
    raise ValueError('the batched engine needs a fixed timestep (dt)')
  c.dt = float(config.dt)
  for a in range(3):
    c.gravity[a] = float(config.gravity[a])
    c.start_pos[a] = float(config.robot_start_pos[a])
  c.motor_torque_limit = float(config.motor_torque_limit)
  c.motor_kp = float(config.motor_kp)
  c.motor_kd = float(config.motor_kd)
  c.linear_damping = float(config.linear_damping)
  c.angular_damping = float(config.angular_damping)
  c.lateral_friction = float(config.lateral_friction)
  c.base_lateral_friction = float(getattr(config, 'base_lateral_friction', 0.5))
  if not c.base_lateral_friction >= 0:
    raise ValueError('base_lateral_friction must be >= 0')
  c.restitution = float(config.restitution)
  c.contact_erp = float(config.contact_erp)
  c.contact_margin = float(config.contact_margin)
  c.joint_limit_margin = float(getattr(config, 'joint_limit_margin', 0.5))
  c.solver_iterations = int(config.solver_iterations)
  tol = getattr(config, 'solver_ulp_tolerance', None)
  c.solver_ulp_tolerance = int(tol) if tol is not None else (DEFAULT_ULP_TOLERANCE_F64 if config.dtype == 'float64' else DEFAULT_ULP_TOLERANCE_F32)
  if c.solver_ulp_tolerance < 0:
    raise ValueError('solver_ulp_tolerance must be >= 0')
  c.solver_residual_threshold = float(getattr(config, 'solver_residual_threshold', 0.0))
  if not c.solver_residual_threshold >= 0:
    raise ValueError('solver_residual_threshold must be >= 0')
  c.settle_steps = int(config.settle_steps)
  q = euler_to_quat(config.robot_start_orientation_euler)
  for a in range(4):
    c.start_quat[a] = q[a]
  if starting_joint_pos is not None:
    for j, name in enumerate(joint_ordering):
      c.settle_targets[j] = float(starting_joint_pos[name])
  # solo8v2vanilla.py:84-85 multiplies by `self._action_space.high`, the float32 bound of the Box built
  # at :170-172: 2 pi reaches the motors as 6.2831854820251465 (float64 arithmetic on a float32 value)
  c.action_scale = float(np.float32(config.max_motor_rotation)) if normalize_actions else 1.0
  c.auto_reset = 1 if config.auto_reset else 0
  def knob(name):   # -1 = the engine chooses
    v = int(getattr(config, name, -1))
    if v < -1:
      raise ValueError('{} must be >= 0, or -1 to let the engine choose'.format(name))
    return v
  c.steps_per_launch = knob('steps_per_launch')
  c.solver_warm_start = float(getattr(config, 'solver_warm_start', 0.0))
  if not 0.0 <= c.solver_warm_start <= 1.0:
    raise ValueError('solver_warm_start must be in [0, 1]')
  if c.solver_warm_start > 0 and not c.solver_residual_threshold > 0:
    raise ValueError('solver_warm_start is an option of the residual-threshold solver: set solver_residual_threshold > 0 '
                     '(pybullet documents 1e-7)')
  c.rollout_streams = knob('rollout_streams')
  c.migrate_steps = knob('migrate_steps')
  return c
