"""ctypes mirror of ``include/solo_engine.h`` (the C-ABI boundary).

Pure layout definitions: no compute, no torch.  Both the HIP engine binding
(``gym_solo_amd.engine``) and the test-only CPU oracle binding (``oracle/solo_oracle.py``)
use these structs, because the robot model and the physics configuration cross the boundary
as DATA (``SoloModel``/``SoloConfig``).
"""
import ctypes as C

ABI_VERSION = 6
NUM_LEGS = 4
NUM_DOF = 8
NUM_JOINTS = 12
NUM_BODIES = 9
NV = 14
MAX_SPHERES = 16
STATE_STRIDE = 32
MAX_OBS = 64
MAX_REWARD_OPS = 32
MAX_TERMS = 4
STATS_SHARDS = 64
STATS_WIDTH = 8

S_POS, S_QUAT, S_Q, S_ANGVEL, S_LINVEL, S_QD, S_RETURN, S_EPLEN, S_SPARE = (
  0, 3, 7, 15, 18, 21, 29, 30, 31)

SRC_EULER, SRC_LINVEL, SRC_ANGVEL, SRC_JPOS, SRC_JVEL, SRC_POS, SRC_QUAT, SRC_ONE = (
  0, 3, 6, 9, 21, 33, 36, 40)
SRC_COUNT = 41

OK = 0
ERR_INVALID_ARG = -1
ERR_HIP = -2
ERR_UNSUPPORTED_MODEL = -3
ERR_NO_PROGRAM = -4
ERR_NO_DEVICE = -5
ERR_INCOMPLETE = -6   # a wave of an earlier migrating launch gave up waiting (internal error; sticky)
AUTO = -1             # steps_per_launch / rollout_streams / migrate_steps: the engine chooses

F32, F64 = 0, 1

OBS_CLIP = 1
OBS_NORMALIZE = 2

(R_CONST, R_UPRIGHT, R_FLAT_TORSO, R_TORSO_HEIGHT, R_HORIZ_SPEED, R_SMALL_CONTROL,
 R_SCALE, R_ADD, R_MUL) = range(9)

T_PERPETUAL, T_TIME, T_CONST = 0, 1, 2

STEP_PHYSICS, STEP_OBS, STEP_REWARD, STEP_DONE, STEP_ALL = 1, 2, 4, 8, 15
STEP_AUTO_RESET = 16  # let a launch without STEP_PHYSICS auto-reset the robots whose `done` fires

PARAM_FRICTION, PARAM_BASE_MASS_SCALE = 0, 1


class SoloModel(C.Structure):
  _fields_ = [
    ('parent', C.c_int32 * NUM_DOF),
    ('joint_origin', (C.c_double * 3) * NUM_DOF),
    ('joint_axis', (C.c_double * 3) * NUM_DOF),
    ('mass', C.c_double * NUM_BODIES),
    ('com', (C.c_double * 3) * NUM_BODIES),
    ('inertia', (C.c_double * 6) * NUM_BODIES),
    ('num_spheres', C.c_int32),
    ('sphere_body', C.c_int32 * MAX_SPHERES),
    ('sphere_center', (C.c_double * 3) * MAX_SPHERES),
    ('sphere_radius', C.c_double * MAX_SPHERES),
    ('dof_to_joint', C.c_int32 * NUM_DOF),
    ('joint_lower', C.c_double * NUM_DOF),
    ('joint_upper', C.c_double * NUM_DOF),
  ]


class SoloConfig(C.Structure):
  _fields_ = [
    ('abi_version', C.c_int32),
    ('dtype', C.c_int32),
    ('dt', C.c_double),
    ('gravity', C.c_double * 3),
    ('motor_torque_limit', C.c_double),
    ('motor_kp', C.c_double),
    ('motor_kd', C.c_double),
    ('linear_damping', C.c_double),
    ('angular_damping', C.c_double),
    ('lateral_friction', C.c_double),
    ('restitution', C.c_double),
    ('contact_erp', C.c_double),
    ('contact_margin', C.c_double),
    ('joint_limit_margin', C.c_double),
    ('solver_iterations', C.c_int32),
    ('settle_steps', C.c_int32),
    ('start_pos', C.c_double * 3),
    ('start_quat', C.c_double * 4),
    ('settle_targets', C.c_double * NUM_JOINTS),
    ('action_scale', C.c_double),
    ('auto_reset', C.c_int32),
    ('steps_per_launch', C.c_int32),
    ('rollout_streams', C.c_int32),
    ('solver_ulp_tolerance', C.c_int32),
    ('solver_residual_threshold', C.c_double),
    ('migrate_steps', C.c_int32),
    ('reserved0', C.c_int32),
    ('solver_warm_start', C.c_double),
    ('base_lateral_friction', C.c_double),
  ]


class SoloObsElem(C.Structure):
  _fields_ = [
    ('src', C.c_int32),
    ('flags', C.c_int32),
    ('scale', C.c_double),
    ('lo', C.c_double),
    ('hi', C.c_double),
    ('nlo', C.c_double),
    ('nhi', C.c_double),
  ]


class SoloRewardInstr(C.Structure):
  _fields_ = [
    ('op', C.c_int32),
    ('pad', C.c_int32),
    ('a', C.c_double),
    ('b', C.c_double),
    ('c', C.c_double),
  ]


class SoloProgram(C.Structure):
  _fields_ = [
    ('num_obs', C.c_int32),
    ('num_reward_ops', C.c_int32),
    ('num_terms', C.c_int32),
    ('pad', C.c_int32),
    ('obs', SoloObsElem * MAX_OBS),
    ('reward', SoloRewardInstr * MAX_REWARD_OPS),
    ('term_kind', C.c_int32 * MAX_TERMS),
    ('term_param', C.c_int32 * MAX_TERMS),
  ]


class SoloTerrain(C.Structure):
  _fields_ = [
    ('nx', C.c_int32),
    ('ny', C.c_int32),
    ('cell', C.c_double),
    ('origin', C.c_double * 2),
    ('heights', C.POINTER(C.c_double)),
  ]


def make_terrain(heights, cell, origin=None):
  """SoloTerrain over a numpy [ny, nx] height array (kept alive on the returned struct); the grid is
  centred on the world origin unless `origin` (x, y of grid point (0, 0)) is given."""
  import numpy as np
  h = np.ascontiguousarray(heights, dtype=np.float64)
  if h.ndim != 2 or h.shape[0] < 2 or h.shape[1] < 2:
    raise ValueError('heights must be a [ny >= 2, nx >= 2] array')
  t = SoloTerrain()
  t.ny, t.nx = h.shape
  t.cell = float(cell)
  if origin is None:
    origin = (-0.5 * (t.nx - 1) * t.cell, -0.5 * (t.ny - 1) * t.cell)
  t.origin[0], t.origin[1] = float(origin[0]), float(origin[1])
  t.heights = h.ctypes.data_as(C.POINTER(C.c_double))
  t._keepalive = h
  return t


class SoloStateView(C.Structure):
  _fields_ = [
    ('num_envs', C.c_int32),
    ('dtype', C.c_int32),
    ('state_stride', C.c_int32),
    ('obs_dim', C.c_int32),
    ('state', C.c_void_p),
    ('snapshot', C.c_void_p),
    ('targets', C.c_void_p),
    ('obs', C.c_void_p),
    ('reward', C.c_void_p),
    ('done', C.c_void_p),
    ('term_count', C.c_void_p),
    ('params', C.c_void_p),
    ('stats', C.c_void_p),
    ('cost', C.c_void_p),
    ('warm', C.c_void_p),
  ]


class SoloLaunchPlan(C.Structure):
  _fields_ = [
    ('steps_per_launch', C.c_int32),
    ('launches', C.c_int32),
    ('slices', C.c_int32),
    ('migrate_steps', C.c_int32),
    ('waves_per_simd', C.c_int32),
    ('resident_robots', C.c_int32),
  ]


# every entry point `include/solo_engine.h` declares: name -> (restype, argtypes)
ENTRY_POINTS = {
  'solo_engine_create': (C.c_int, [C.POINTER(SoloConfig), C.POINTER(SoloModel), C.c_int32,
                                   C.c_int32, C.POINTER(C.c_void_p)]),
  'solo_engine_destroy': (C.c_int, [C.c_void_p]),
  'solo_engine_set_program': (C.c_int, [C.c_void_p, C.POINTER(SoloProgram)]),
  'solo_engine_reset': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
  'solo_engine_settle': (C.c_int, [C.c_void_p, C.c_void_p]),
  'solo_engine_set_targets': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
  'solo_engine_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
  'solo_engine_rollout': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p]),
  'solo_engine_rollout_record': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
  'solo_engine_get_view': (C.c_int, [C.c_void_p, C.POINTER(SoloStateView)]),
  'solo_engine_set_params': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
  'solo_engine_set_terrain': (C.c_int, [C.c_void_p, C.POINTER(SoloTerrain), C.c_void_p]),
  'solo_engine_set_order': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
  'solo_engine_kernel_name': (C.c_char_p, [C.c_void_p]),
  'solo_engine_time_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32,
                                      C.c_void_p, C.POINTER(C.c_double)]),
  'solo_engine_plan': (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(SoloLaunchPlan)]),
  'solo_engine_time_rollout': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_double)]),
  'solo_engine_reserve': (C.c_int, [C.c_void_p, C.c_int32, C.c_uint32]),
  'solo_engine_last_error': (C.c_char_p, [C.c_void_p]),
  'solo_last_create_error': (C.c_char_p, []),
  'solo_abi_version': (C.c_int, []),
}


def bind(lib):
  """Attach restype/argtypes for every declared entry point; raises AttributeError if the
  library does not export one of them."""
  for name, (res, args) in ENTRY_POINTS.items():
    fn = getattr(lib, name)
    fn.restype = res
    fn.argtypes = args
  return lib
