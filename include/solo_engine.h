/* solo_engine.h — C-ABI of the MI355X batched Solo8 physics engine.
 *
 * This is the drop-in boundary for the ONE hot path of WPI-MMR/gym_solo:
 * the per-step pybullet call sequence issued by Solo8VanillaEnv.step()
 *   client.setJointMotorControlArray(...)   gym_solo/envs/solo8v2vanilla.py:87-90
 *   client.stepSimulation()                 gym_solo/envs/solo8v2vanilla.py:91
 *   client.getBasePositionAndOrientation    gym_solo/core/obs.py:268, rewards.py:232,264,370
 *   client.getBaseVelocity                  gym_solo/core/obs.py:273, rewards.py:334
 *   client.getJointState                    gym_solo/core/obs.py:354, rewards.py:298
 * plus the obs/reward/termination reductions (obs.py:130-159, rewards.py:104-118,
 * termination.py:38-50) and the reset/settle loop (solo8v2vanilla.py:104-143),
 * for N independent robots at once.
 *
 * The reference binds pybullet through `pybullet_utils.bullet_client.BulletClient`
 * (solo8_base_env.py:34-35).  A maintainer replaces that object by a ctypes stub
 * over this header; no torch / C++ type crosses the boundary:
 * plain pointers (device pointers are `void*`), sizes and PODs only.
 *
 * Conventions
 *  - return value: 0 = SOLO_OK, negative = SoloStatus error; text via
 *    solo_engine_last_error().
 *  - one engine handle per GPU / process; NOT thread-safe per handle; all work
 *    is enqueued on the caller-provided HIP stream (NULL = default stream), no
 *    hidden device synchronisation except where stated.
 *  - device buffers are owned by the engine and exposed zero-copy through
 *    SoloStateView; pointers passed IN (actions, masks, params) are borrowed.
 *  - `real` below means float (dtype SOLO_F32) or double (SOLO_F64) as chosen at
 *    create time; every device buffer of the engine uses that type.
 */
#ifndef SOLO_ENGINE_H_
#define SOLO_ENGINE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOLO_ABI_VERSION 6  /* 2: SOLO_STEP_AUTO_RESET; query-only launches never auto-reset; reset restores the motor targets
                               3: SoloConfig::solver_residual_threshold
                               4: SoloConfig::migrate_steps, SoloConfig::solver_warm_start, SoloStateView::warm
                               5: -1 = "the engine chooses" for steps_per_launch / rollout_streams / migrate_steps (the measured
                                  launch policy lives in the engine: solo_engine_plan reports it), solo_engine_time_rollout,
                                  SOLO_ERR_INCOMPLETE
                               6: SoloConfig::base_lateral_friction (the base link keeps its own friction: the reference's
                                  changeDynamics loop never reaches link -1), solo_engine_reserve */

/* ---- fixed Solo8 dimensions -------------------------------------------- */
#define SOLO_NUM_LEGS 4
#define SOLO_NUM_DOF 8        /* actuated revolute joints (HFE, KFE per leg)          */
#define SOLO_NUM_JOINTS 12    /* pybullet joint count incl. 4 fixed ANKLE joints:      */
                              /* gym_solo/envs/test_solo8v2vanilla.py:57               */
#define SOLO_NUM_BODIES 9     /* base + 8 moving links (feet are welded to lower legs) */
#define SOLO_NV 14            /* velocity dofs: 6 (floating base) + 8                  */
#define SOLO_MAX_SPHERES 16   /* collision primitives (sphere vs ground)               */
#define SOLO_STATE_STRIDE 32  /* reals per env record (one 128-B line in f32)          */
#define SOLO_MAX_OBS 64       /* observation elements per env                          */
#define SOLO_MAX_REWARD_OPS 32
#define SOLO_MAX_TERMS 4
#define SOLO_STATS_SHARDS 64  /* episodic statistics are accumulated in 64 rows (sum them) */
#define SOLO_STATS_WIDTH 8

/* env record layout (reals), AoS so that one wavefront reads its robot's whole
 * state as ONE coalesced 128-B (f32) / 256-B (f64) line: */
#define SOLO_S_POS 0      /* base CoM position, world            [3] */
#define SOLO_S_QUAT 3     /* base orientation xyzw (pybullet)    [4] */
#define SOLO_S_Q 7        /* joint angles, dof order             [8] */
#define SOLO_S_ANGVEL 15  /* base angular velocity, world frame  [3] */
#define SOLO_S_LINVEL 18  /* base linear velocity, world frame   [3] */
#define SOLO_S_QD 21      /* joint velocities                    [8] */
#define SOLO_S_RETURN 29  /* episodic return accumulator             */
#define SOLO_S_EPLEN 30   /* episode length accumulator              */
#define SOLO_S_SPARE 31

typedef enum SoloStatus {
  SOLO_OK = 0,
  SOLO_ERR_INVALID_ARG = -1, /* -> ValueError in the Python facade */
  SOLO_ERR_HIP = -2,         /* -> RuntimeError                    */
  SOLO_ERR_UNSUPPORTED_MODEL = -3,
  SOLO_ERR_NO_PROGRAM = -4,  /* step() without a reward / obs / termination registered:
                                mirrors the ValueErrors of rewards.py:115-116, obs.py:138-139,
                                termination.py:43-44 */
  SOLO_ERR_NO_DEVICE = -5,
  SOLO_ERR_INCOMPLETE = -6   /* a wave of an EARLIER launch with robot migration gave up waiting for its robot (a bounded
                                wait that a correct queue never exhausts): some robots were not stepped through that
                                launch.  Sticky: every later call on the handle that takes the engine (everything but
                                destroy, last_error and kernel_name) returns it.  An internal error - never observed -
                                surfaced instead of returned as SOLO_OK; the count is slot 6 of the statistics. */
} SoloStatus;

typedef enum SoloDType { SOLO_F32 = 0, SOLO_F64 = 1 } SoloDType;

/* ---- robot model (DATA, not code: both the engine and the CPU oracle take it) */
typedef struct SoloModel {
  /* body 0 = base; body 1+j = link moved by dof j.  dof order:
   * FL_HFE, FL_KFE, FR_HFE, FR_KFE, HL_HFE, HL_KFE, HR_HFE, HR_KFE. */
  int32_t parent[SOLO_NUM_DOF];            /* parent BODY index of dof j's link         */
  double joint_origin[SOLO_NUM_DOF][3];    /* joint frame origin, in parent body frame  */
  double joint_axis[SOLO_NUM_DOF][3];      /* unit axis (Solo8: all +y)                 */
  double mass[SOLO_NUM_BODIES];
  double com[SOLO_NUM_BODIES][3];          /* in body frame; base com must be 0         */
  double inertia[SOLO_NUM_BODIES][6];      /* about com, body axes: xx yy zz xy xz yz   */
  int32_t num_spheres;
  int32_t sphere_body[SOLO_MAX_SPHERES];
  double sphere_center[SOLO_MAX_SPHERES][3]; /* in body frame */
  double sphere_radius[SOLO_MAX_SPHERES];
  /* dof j <-> pybullet joint index (for getJointState facade), and the 12 names */
  int32_t dof_to_joint[SOLO_NUM_DOF];
  /* URDF joint limits [rad] (the reference's getJointInfo fixture pins -10 / +10,
   * gym_solo/core/test_obs_observations.py:123-162 columns 8-9): unilateral rows next to the motors
   * ([recalled] btMultiBodyJointLimitConstraint).  lower < upper required. */
  double joint_lower[SOLO_NUM_DOF];
  double joint_upper[SOLO_NUM_DOF];
} SoloModel;

/* ---- physics / env configuration (gym_solo/core/configs.py:8-38) -------- */
typedef struct SoloConfig {
  int32_t abi_version;       /* = SOLO_ABI_VERSION */
  int32_t dtype;             /* SoloDType */
  double dt;                 /* configs.py:10  (1e-3), one substep: solo8_base_env.py:39-41 */
  double gravity[3];         /* configs.py:17 */
  double motor_torque_limit; /* configs.py:12  -> impulse clamp +-limit*dt per step */
  double motor_kp;           /* pybullet POSITION_CONTROL default positionGain 0.1 [recalled] */
  double motor_kd;           /* pybullet POSITION_CONTROL default velocityGain 1.0 [recalled] */
  double linear_damping;     /* configs.py:21 */
  double angular_damping;    /* configs.py:22 */
  double lateral_friction;   /* configs.py:24 (x plane friction 1.0): the collision spheres of links 0..11 - the legs.  The
                                reference's changeDynamics loop runs over range(getNumJoints) (solo8v2vanilla.py:157-163) and so
                                never reaches the base link (-1): see base_lateral_friction */
  double restitution;        /* configs.py:23, passed to changeDynamics for links 0..11 (solo8v2vanilla.py:158-163).  In [0, 1];
                                WITHOUT EFFECT here, as in the reference: Bullet gives a contact the PRODUCT of its two bodies'
                                restitutions ([recalled] btManifoldResult::calculateCombinedRestitution), and the ground the
                                reference loads - pybullet_data's plane.urdf, solo8_base_env.py:47 - declares none (0) */
  double contact_erp;        /* penetration recovery rate (Bullet erp2 0.2 [recalled]) */
  double contact_margin;     /* spheres closer than this to the ground create rows */
  double joint_limit_margin; /* a joint closer than this [rad] to one of its limits gets that limit's row
                                (a speculative unilateral row: exact as long as |qd| dt < margin) */
  int32_t solver_iterations; /* Bullet default 50 [recalled] */
  int32_t settle_steps;      /* solo8v2vanilla.py:130 (500) */
  double start_pos[3];       /* configs.py:15 */
  double start_quat[4];      /* from configs.py:16 euler */
  double settle_targets[SOLO_NUM_JOINTS]; /* solo8v2vanilla.py:21-34, pybullet joint order */
  double action_scale;       /* normalize_actions ? max_motor_rotation : 1 (solo8v2vanilla.py:84-85) */
  int32_t auto_reset;        /* 1: envs whose `done` fires are restored from the snapshot in-kernel */
  int32_t steps_per_launch;  /* rollouts / settle fuse this many consecutive env steps of each robot
                                into one kernel launch (state stays in LDS); 0 or 1 = one step;
                                -1 = the engine chooses (a rollout of K steps: launches of min(K, 250) steps) */
  int32_t rollout_streams;   /* rollouts cut the batch into this many slices that advance as
                                independent launch chains on internal HIP streams (0/1 = off);
                                -1 = the engine chooses (two slices when a rollout takes more than one launch) */
  int32_t solver_ulp_tolerance; /* k: a Gauss-Seidel row whose clamped candidate differs from its impulse by
                                at most k half-ulps relative to the impulse (|d| <= k * 2^-24 |lam| in f32,
                                k * 2^-53 |lam| in f64) is left untouched, and a sweep that changes no row
                                ends the iteration early.  0 = only an exact fixed point ends it (then
                                identical to always running solver_iterations sweeps); 2 also stops last-bit limit
                                cycles, which otherwise keep ~3% of the robots iterating to the cap.  Host defaults
                                (gym_solo_amd/core/configs.py): 2 in f32, 512 in f64 - 5.7e-14 relative, half of the
                                rounding noise between two f64 formulations of one step; measured in round 6: rest
                                behaviour and parity against 50 plain sweeps unchanged, 12 % fewer sweeps.
                                Negative values are rejected. */
  double solver_residual_threshold; /* pybullet's solverResidualThreshold ([recalled] documented default 1e-7, set in
                                PhysicsServerCommandProcessor::createEmptyDynamicsWorld; the reference never changes it;
                                the host default here is 0 = off - gym_solo_amd/core/configs.py says why):
                                the iteration ends after the first sweep in which max over the rows of
                                (delta impulse x A_rr)^2 - the squared velocity-level change, [recalled]
                                btMultiBodyConstraintSolver::solveSingleIteration / resolveSingleConstraintRowGeneric
                                returning deltaImpulse / jacDiagABInv - is <= this value.  0 = never (every sweep that
                                still changes a row runs, up to solver_iterations).  Negative values are rejected. */
  int32_t migrate_steps;     /* c > 0: a fused launch of more than c steps hands its robots from wave to wave every c
                                steps through a work queue in device memory (the robot's 256-B record travels; any idle
                                wave continues any robot), so that the launch ends when the WORK is done and not when
                                the unluckiest SIMD's robots are.  Scheduling only: results are bit-identical.  0 = off
                                (one wave steps one robot through the whole launch).  -1 = the engine chooses: off while
                                every robot of a launch has a wave slot of its own (4096 robots: four waves on each of
                                the chip's 1024 SIMDs, in both precisions since round 5), else - in f64 - two chunks per
                                launch (chunks of 25 steps in a rollout of several launches); f32 never (measured slower) -
                                solo_engine_plan reports it. */
  int32_t reserved0;         /* (padding; must be 0) */
  double solver_warm_start;  /* f in (0, 1]: the Gauss-Seidel iteration of a step STARTS from f x the impulses the previous
                                step ended with (every row: motors, joint limits, contacts - clamped to the row's bounds
                                of this step, the friction rows to mu x their contact's starting normal impulse; a row
                                that was not live in the previous step starts at 0; a reset / auto-reset / restored robot
                                starts at 0).  [recalled] Bullet warm-starts its rigid-body contacts with
                                m_warmstartingFactor 0.85; whether the btMultiBody path the reference runs on does is
                                not known here.  An OPT-IN, and only together with solver_residual_threshold > 0 (the pair
                                is what makes pybullet's early exit leave a resting robot at rest: DESIGN.md section 4).
                                0 = off (every step starts from zero impulses).  The cache - SoloStateView::warm, 64 reals
                                per robot, the step kernel's lane layout - costs 2 x 64 reals of memory traffic per
                                env-step, reported separately from the path's algorithmic bytes. */
  double base_lateral_friction; /* friction of the BASE link's collision spheres (model spheres attached to body 0).  The
                                reference sets lateralFriction with changeDynamics for links 0 .. 11 only
                                (solo8v2vanilla.py:157-163: `for joint in range(joint_cnt)`), so the base keeps what loadURDF
                                gave it: [recalled] pybullet's default 0.5 (x plane.urdf's 1.0).  Neither
                                SoloConfig::lateral_friction nor solo_engine_set_params(0, ...) touches it: with the
                                reference's 0.5 the two coincide, with any other leg friction (BASELINE configs[3]'s
                                per-robot values) the belly keeps 0.5.  Host default 0.5; negative values are rejected. */
} SoloConfig;

/* ---- fused observation / reward / termination programs ------------------ */
/* Source vector the obs program indexes (per env, computed in-kernel):
 *   [0..2] euler xyz  [3..5] base lin vel  [6..8] base ang vel
 *   [9..20] joint angle by pybullet joint index (fixed joints = 0)
 *   [21..32] joint velocity by pybullet joint index
 *   [33..35] base position  [36..39] quaternion xyzw  [40] constant 1.0 */
#define SOLO_SRC_EULER 0
#define SOLO_SRC_LINVEL 3
#define SOLO_SRC_ANGVEL 6
#define SOLO_SRC_JPOS 9
#define SOLO_SRC_JVEL 21
#define SOLO_SRC_POS 33
#define SOLO_SRC_QUAT 36
#define SOLO_SRC_ONE 40
#define SOLO_SRC_COUNT 41

typedef struct SoloObsElem {
  int32_t src;      /* index into the source vector */
  int32_t flags;    /* bit0: clip to [lo,hi]; bit1: normalise 2(a-nlo)/(nhi-nlo)-1 (obs.py:149-152) */
  double scale;     /* e.g. 180/pi for degrees (obs.py:277-279, 358) */
  double lo, hi;    /* clip bounds (obs.py:282, 361) */
  double nlo, nhi;  /* normalisation bounds: the float32 Box bounds widened to double */
} SoloObsElem;

/* reward program: postfix (RPN) over a small value stack */
typedef enum SoloRewardOp {
  SOLO_R_CONST = 0,        /* push a                                              */
  SOLO_R_UPRIGHT = 1,      /* push (-pi/2)*pitch/(-pi/2)^2      rewards.py:221-234 */
  SOLO_R_FLAT_TORSO = 2,   /* push tol(sqrt(tx^2+ty^2),(-a,a),b) rewards.py:256-269 */
  SOLO_R_TORSO_HEIGHT = 3, /* push tol(z,(a-b,a+b),c)            rewards.py:362-373 */
  SOLO_R_HORIZ_SPEED = 4,  /* push tol(|vxy|,(a-b,a+b),c)        rewards.py:326-338 */
  SOLO_R_SMALL_CONTROL = 5,/* push tol(mean12|qd|,(0,0),a)       rewards.py:290-301 */
  SOLO_R_SCALE = 6,        /* top *= a                                            */
  SOLO_R_ADD = 7,          /* pop b, pop a, push a+b                              */
  SOLO_R_MUL = 8           /* pop b, pop a, push a*b                              */
} SoloRewardOp;

typedef struct SoloRewardInstr {
  int32_t op;
  int32_t pad;
  double a, b, c;
} SoloRewardInstr;

typedef enum SoloTermKind {
  SOLO_T_PERPETUAL = 0,  /* termination.py:86-97  */
  SOLO_T_TIME = 1,       /* termination.py:59-83  */
  SOLO_T_CONST = 2       /* testing.py DummyTermination: fixed flag */
} SoloTermKind;

typedef struct SoloProgram {
  int32_t num_obs;
  int32_t num_reward_ops;
  int32_t num_terms;
  int32_t pad;
  SoloObsElem obs[SOLO_MAX_OBS];
  SoloRewardInstr reward[SOLO_MAX_REWARD_OPS];
  int32_t term_kind[SOLO_MAX_TERMS];
  int32_t term_param[SOLO_MAX_TERMS]; /* max_step_delta / flag */
} SoloProgram;

/* ---- ground: the reference loads pybullet_data's flat `plane.urdf` (solo8_base_env.py:47);
 * BASELINE configs[4] asks for inclined / stair terrain.  A heightfield z = h(x, y) on a regular
 * grid, bilinearly interpolated (clamped to the border outside the grid); every collision sphere
 * collides with the tangent plane of the terrain under its centre. ------------------------- */
typedef struct SoloTerrain {
  int32_t nx, ny;          /* grid points along x and y (>= 2 each)                       */
  double cell;             /* grid spacing [m]                                            */
  double origin[2];        /* world (x, y) of grid point (0, 0)                           */
  const double* heights;   /* HOST pointer, [ny][nx] row-major (y major), copied by the call */
} SoloTerrain;

/* ---- zero-copy view of engine-owned device buffers ----------------------- */
typedef struct SoloStateView {
  int32_t num_envs;
  int32_t dtype;        /* SoloDType of every `real` buffer               */
  int32_t state_stride; /* = SOLO_STATE_STRIDE                            */
  int32_t obs_dim;      /* D of the current program (0 if none)           */
  void* state;          /* real  [N][SOLO_STATE_STRIDE]                   */
  void* snapshot;       /* real  [N][SOLO_STATE_STRIDE] post-settle state */
  void* targets;        /* real  [N][12] last motor targets (radians)     */
  void* obs;            /* real  [N][D]                                   */
  void* reward;         /* real  [N]                                      */
  void* done;           /* uint8 [N]                                      */
  void* term_count;     /* int32 [N][SOLO_MAX_TERMS]                      */
  void* params;         /* real  [N][4]: lateral friction, base-mass scale, 2 spare */
  void* stats;          /* double[SOLO_STATS_SHARDS][8], sum over the shard axis: sum return, sum return^2,
                           episodes, sum length, (unused), diverged robots restored, waves of migrating launches that
                           gave up waiting (row 0 only; non-zero = SOLO_ERR_INCOMPLETE), 1 spare */
  void* cost;           /* int32 [N]: Gauss-Seidel sweeps each robot ran in the LAST launch that stepped it
                           (its cost is persistent within an episode): input of solo_engine_set_order */
  void* warm;           /* real  [N][64]: the impulses every robot's last step ended with, one per constraint row in the
                           step kernel's lane layout (solo_kernel_params.h) - the warm-start cache of
                           SoloConfig::solver_warm_start; all zero while that is off */
} SoloStateView;

typedef struct SoloEngine SoloEngine;

/* flags of solo_engine_step */
#define SOLO_STEP_PHYSICS 1u  /* A3+A4: motors + stepSimulation */
#define SOLO_STEP_OBS 2u      /* A5-A7  */
#define SOLO_STEP_REWARD 4u   /* A8-A11 */
#define SOLO_STEP_DONE 8u     /* A12 (+ auto-reset when configured) */
#define SOLO_STEP_ALL 15u
/* The in-kernel auto-reset (cfg.auto_reset) acts in launches that carry SOLO_STEP_PHYSICS: a
 * query-only launch (TerminationFactory.is_terminated() outside step(), termination.py:38-50)
 * never mutates the simulation.  A caller that evaluates `done` in a launch of its own AFTER the
 * physics launch (a Python-side observation / reward in between) adds this bit to let that launch
 * restore the robots whose `done` fires. */
#define SOLO_STEP_AUTO_RESET 16u

/* BulletClient(connection_mode) + setGravity + setPhysicsEngineParameter + loadURDF x2
 * (solo8_base_env.py:34-48).  Allocates device buffers, uploads the model, places every
 * robot at start_pos and runs the settle loop (solo8v2vanilla.py:124-136) to build the
 * reset snapshot.  Synchronises the device once. */
int solo_engine_create(const SoloConfig* cfg, const SoloModel* model, int32_t num_envs,
                       int32_t device_id, SoloEngine** out);
/* client.disconnect()  (solo8_base_env.py:150-152) */
int solo_engine_destroy(SoloEngine* eng);
/* register_observation / register_reward / register_termination, compiled
 * (obs.py:109-128, rewards.py:91-102, termination.py:28-36) */
int solo_engine_set_program(SoloEngine* eng, const SoloProgram* prog);
/* resetSimulation + reload + settle (solo8v2vanilla.py:104-143): restores the snapshot for
 * envs with mask[i] != 0 (mask NULL = all) and clears their termination counters. */
int solo_engine_reset(SoloEngine* eng, const uint8_t* mask_dev, void* stream);
/* re-run the settle loop from the start pose for every env and refresh the snapshot
 * (needed after solo_engine_set_params). Synchronises the stream. */
int solo_engine_settle(SoloEngine* eng, void* stream);
/* setJointMotorControlArray(..., POSITION_CONTROL, targetPositions=a, forces=limit)
 * (solo8v2vanilla.py:87-90).  actions_dev: real [N][12] in pybullet joint order
 * (entries 2,5,8,11 = fixed ANKLE joints are ignored); multiplied by action_scale. */
int solo_engine_set_targets(SoloEngine* eng, const void* actions_dev, void* stream);
/* One env step for all N robots (solo8v2vanilla.py:72-102).  actions_dev may be NULL
 * (keep the last targets).  flags: SOLO_STEP_*.  One kernel launch. */
int solo_engine_step(SoloEngine* eng, const void* actions_dev, uint32_t flags, void* stream);
/* K consecutive env steps with per-step actions real [K][N][12] (open-loop rollout, used by
 * bench.py): ceil(K / steps_per_launch) launches enqueued back-to-back; the view's obs / reward /
 * done hold the LAST step's values afterwards.  With cfg.rollout_streams = G > 1 the batch is cut
 * into G slices that advance as independent launch chains on G internal HIP streams, forked from
 * / joined into `stream` (robots are independent: one slice's launch boundary and tail overlap
 * the other slices' work). */
int solo_engine_rollout(SoloEngine* eng, const void* actions_dev, int32_t num_steps,
                        uint32_t flags, void* stream);
/* Same, but every step's outputs are kept: obs_out real [K][N][D], reward_out real [K][N],
 * done_out uint8 [K][N] (caller-owned device buffers; a NULL pointer = that output is not recorded).
 * The engine's view holds the last step's outputs afterwards in either case.  This is what an RL
 * rollout collector reads. */
int solo_engine_rollout_record(SoloEngine* eng, const void* actions_dev, int32_t num_steps,
                               uint32_t flags, void* obs_out, void* reward_out, void* done_out,
                               void* stream);
int solo_engine_get_view(SoloEngine* eng, SoloStateView* out);
/* changeDynamics(lateralFriction=...) per env + base-mass randomisation
 * (solo8v2vanilla.py:158-163; BASELINE config 4).  which: 0 = friction, 1 = base mass scale.
 * per_env_dev: real [N]. */
int solo_engine_set_params(SoloEngine* eng, int32_t which, const void* per_env_dev, void* stream);
/* Replaces loadURDF('plane.urdf') (solo8_base_env.py:47): NULL = the flat plane z = 0.  Re-runs the
 * settle loop (the reset snapshot depends on the ground).  Synchronises the device. */
int solo_engine_set_terrain(SoloEngine* eng, const SoloTerrain* terrain, void* stream);
/* Launch order of the robots (cost-balanced scheduling): order_dev int32 [N], a permutation of 0..N-1
 * that maps workgroup b of a launch to robot order[b]; NULL = the engine's own map (each of the 8 XCDs of the
 * chip steps a contiguous eighth of the launch's robots, so that neighbouring robots' rows of the [K][N][.]
 * arrays meet in one L2: HBM traffic, not results).  With cfg.rollout_streams = G
 * the positions [N g / G, N (g+1) / G) must hold a permutation of the same robot range (every slice
 * keeps its robots).  Results do not depend on the order (robots are independent); it matters when
 * N exceeds the 4096 resident waves of the chip: dispatching the costliest robots first (view.cost,
 * descending) keeps the tail of a launch short.  Copied into an engine-owned buffer.  The table is
 * VALIDATED on upload (one device-to-host copy, synchronises `stream`): anything but such a permutation
 * returns SOLO_ERR_INVALID_ARG and leaves the previous order in force. */
int solo_engine_set_order(SoloEngine* eng, const int32_t* order_dev, void* stream);
/* name of the dominant kernel (for rocprof cross-checks) and its last launch geometry */
const char* solo_engine_kernel_name(SoloEngine* eng);
/* Times a rollout of reps * steps_per_launch steps - the CONFIGURED steps per launch; one step per launch when that is
 * left to the engine (solo_engine_time_rollout times a rollout with the engine's own geometry) - as solo_engine_rollout
 * runs it (same slicing, same fused launches; the step kernel, its output epilogue included) with hipEvents
 * recorded ON THE STREAMS THE KERNELS ARE LAUNCHED ON (every slice's internal stream when
 * rollout_streams > 1, else `stream`) and returns the mean milliseconds per LAUNCH over all slices;
 * one launch covers N / max(1, rollout_streams) robots x steps_per_launch steps.
 * actions_dev: real [reps * steps_per_launch][N][12] or NULL. */
int solo_engine_time_step(SoloEngine* eng, const void* actions_dev, uint32_t flags,
                          int32_t reps, void* stream, double* ms_per_launch);
/* The launch geometry a rollout of num_steps steps runs with (what -1 = "the engine chooses" resolved to, or the
 * configured values): the measured launch policy is the engine's, not the caller's. */
typedef struct SoloLaunchPlan {
  int32_t steps_per_launch;  /* env steps fused into one launch (the last launch of a slice may be shorter) */
  int32_t launches;          /* launches per slice */
  int32_t slices;            /* independent launch chains (batch slices on internal HIP streams) */
  int32_t migrate_steps;     /* steps per migration chunk of a launch, 0 = no robot migration */
  int32_t waves_per_simd;    /* resident step-kernel waves per SIMD in this precision (4) */
  int32_t resident_robots;   /* robots with a wave slot of their own on this device: waves_per_simd x SIMDs */
} SoloLaunchPlan;
int solo_engine_plan(SoloEngine* eng, int32_t num_steps, SoloLaunchPlan* out);
/* LAZY SCRATCH.  The per-launch record scratch of fused launches ([N][steps_per_launch][32] reals: 0.5 GB for 8192 robots x
 * 250 steps in f64) and the robot-migration queues are sized for the largest rollout geometry seen so far: the FIRST
 * rollout / step / time_* call of a larger geometry synchronises the DEVICE, frees and re-allocates them (never shrinks) -
 * the one hidden synchronisation of the otherwise stream-ordered calls, not legal inside a HIP stream capture, and the
 * place where an out-of-memory surfaces (SOLO_ERR_HIP) after create.  solo_engine_reserve does that work NOW for rollouts
 * of num_steps steps with these flags (and every shorter one under the same configuration): latency-sensitive or
 * graph-capturing callers call it once after solo_engine_set_program; later calls within the reserved geometry allocate
 * nothing and synchronise nothing.  Synchronises the device when it grows something. */
int solo_engine_reserve(SoloEngine* eng, int32_t num_steps, uint32_t flags);
/* Times ONE rollout of num_steps steps exactly as solo_engine_rollout_record runs it (solo_engine_plan's geometry; the
 * output buffers as there: NULL = that output is not recorded) with hipEvents recorded on the streams the kernels are
 * launched on; returns the mean milliseconds per LAUNCH over all slices and launches.  actions_dev: real [num_steps][N][12]. */
int solo_engine_time_rollout(SoloEngine* eng, const void* actions_dev, int32_t num_steps, uint32_t flags,
                             void* obs_out, void* reward_out, void* done_out, void* stream, double* ms_per_launch);
const char* solo_engine_last_error(SoloEngine* eng);
/* library-level: last error of a failed create (eng == NULL) */
const char* solo_last_create_error(void);
int solo_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SOLO_ENGINE_H_ */
