// solo_kernel_params.h — parameter block of the fused step kernel and its host-side packing.
//
// The block is wave-uniform: the kernel receives a pointer to it in global memory and the
// compiler turns the uniform reads into scalar (SMEM) loads.  Only the per-leg and per-row
// tables are indexed by lane.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>

#include "../../include/solo_engine.h"

namespace solo {

// lane layout of one wavefront = one robot.  lane = 16*leg + k:
//   k 0,1      motor rows of the leg's HFE / KFE
//   k 2+3j..4+3j  model sphere 4*leg+j (j = 0..3): normal, tangent x, tangent y.  Spheres 4l and
//              4l+1 ride on leg l (knee, foot), 4l+2 and 4l+3 on the base.  Model sphere order ==
//              ascending lane order, so "the touching spheres in solve order" is simply the set
//              bits of the touching-lanes ballot, lowest first.
//   k 14,15    joint-limit rows of the leg's HFE / KFE (the nearer limit; live only close to it)
enum RowType : int32_t { ROW_IDLE = 0, ROW_MOTOR = 1, ROW_NORMAL = 2, ROW_TAN1 = 3, ROW_TAN2 = 4, ROW_LIMIT = 5 };
enum BodyKind : int32_t { BODY_BASE = 0, BODY_UPPER = 1, BODY_LOWER = 2 };

__host__ __device__ constexpr int motor_lane(int dof) { return 16 * (dof >> 1) + (dof & 1); }
__host__ __device__ constexpr int sphere_lane(int s) { return 16 * (s >> 2) + 2 + 3 * (s & 3); }

// The 27 base-level terms a leg contributes (lower triangle of the 6 x 6 Schur complement, row by row, then the right-hand
// side), as LANE e of the wave evaluates them from what each leg has posted in LDS (solo_step_kernel.h, f64):
//   term = sign * P[b] - A1[i] * A1[j] - A2[i] * A2[j]
// P = [IO 0..5 | m c 6..8 | m_leg 9 | 0 10 | -N 11..13 | -F 14..16], A1 = [W1 0..5 | -e1], A2 = [W2 0..5 | -e2].
// Entry: b | negate << 5 | i << 6 | j << 9.
__host__ __device__ constexpr int leg_sum_entry(int e) {
  constexpr int Z = 10, N = 32;   // (Z: the posted zero; N: negate)
  constexpr int b[27] = {0, 3, 1, 4, 5, 2,                 // rows 0..2: the composite inertia about the base origin (xx yy zz xy xz yz)
                         Z, 8, 7 | N, 9,                    // row 3: [0, m c_z, -m c_y | m]
                         8 | N, Z, 6, Z, 9,                 // row 4: [-m c_z, 0, m c_x | 0, m]
                         7, 6 | N, Z, Z, Z, 9,              // row 5: [m c_y, -m c_x, 0 | 0, 0, m]
                         11, 12, 13, 14, 15, 16};           // the right-hand side
  constexpr int i[27] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 0, 1, 2, 3, 4, 5};
  constexpr int j[27] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 5, 6, 6, 6, 6, 6, 6};
  return b[e] | (i[e] << 6) | (j[e] << 9);
}

template <typename T>
struct LegConst {
  T hip[3];     // HFE joint origin in base frame
  T knee[3];    // KFE joint origin in upper-leg frame
  T link[2][10];  // [0] upper leg, [1] lower leg + welded foot: mass, com[3] (link frame),
                  // inertia about com [6] (xx yy zz xy xz yz)
  T limit[2][2];  // [HFE, KFE][lower, upper] URDF joint limits [rad]
};

template <typename T>
struct RowConst {
  int32_t type;   // RowType
  int32_t body;   // BodyKind of the sphere's body
  int32_t dof;    // motor rows: dof index (0..7) ; contact rows: model sphere index
  int32_t pad;
  T center[3];    // sphere centre in its body frame
  T radius;
};

template <typename T>
struct ObsElemK {
  int32_t src, flags;
  T scale, lo, hi, nscale, noff;  // normalise: 2*(a - nlo)/(nhi - nlo) - 1 (obs.py:152) as a * nscale + noff
};

template <typename T>
struct RewardInstrK {
  int32_t op;
  int32_t src;  // three-address form of the postfix program: src0 | src1 << 8 = indices of the
                // instructions whose results this one consumes (SCALE: src0; ADD / MUL: both)
  T a, b, c;
  T d;  // tolerance leaves: gaussian scale / margin (rewards.py:427), 0 when the margin is 0
};

// The scalars one env step reads.  The step kernel stages this block into LDS once per launch and
// every step reads it from there (LDS broadcasts): read straight from global memory through the
// per-step re-derived parameter pointer they were ~15 flat loads per step, each followed by an
// exposed s_waitcnt vmcnt(0); kept in scalar registers across the fused step loop they spill.
template <typename T>
struct StepConst {
  T dt, inv_dt;
  T gravity[3];
  T kp_over_dt, one_minus_kd, motor_impulse;
  T lin_damp, ang_damp;
  T erp_over_dt, margin;
  T limit_margin;  // SoloConfig::joint_limit_margin
  T resid_thr;     // SoloConfig::solver_residual_threshold (squared velocity-level change; 0 = off)
  T warm_factor;   // SoloConfig::solver_warm_start (0 = every step starts from zero impulses)
  T action_scale;
  // heightfield ground (SoloTerrain): 1/cell, origin; grid size below; heights live in KBuffers::terrain
  T terr_inv_cell, terr_ox, terr_oy;
  T base_mass, base_I[6];
  T settle_tgt[SOLO_NUM_JOINTS];  // motor targets a reset leaves behind [rad] (solo8v2vanilla.py:21-34,127-136)
  int32_t iterations, auto_reset;
  int32_t ulp_tol;                // SoloConfig::solver_ulp_tolerance
  int32_t terr_nx, terr_ny;
  int32_t num_terms, num_obs, num_reward_ops;
  int32_t term_kind[SOLO_MAX_TERMS];
  int32_t term_param[SOLO_MAX_TERMS];
};

template <typename T>
struct KParams {
  StepConst<T> c;
  LegConst<T> leg[4];
  RowConst<T> row[64];
  ObsElemK<T> obs[SOLO_MAX_OBS];
  RewardInstrK<T> reward[SOLO_MAX_REWARD_OPS];
  T mu_base;   // SoloConfig::base_lateral_friction: the friction rows of the BASE link's spheres (the robot's own coefficient -
               // params[e][0] - is the legs').  Not part of StepConst: the f64 kernel's LDS is exactly eight allocation granules
               // (10240 B) - the prologue parks it in s_keep[29] next to the robot's own coefficient.
};

// Workgroup -> robot map of a launch with no explicit order (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement":
// workgroups are dealt round-robin over the 8 XCDs - b and b + 8 share one - and every XCD has its own L2).  The
// robots an XCD steps are made a CONTIGUOUS range of the batch: the per-step rows the waves write ([step][robot][.]
// arrays: 84-B observation rows, 48-B action rows - two or three robots per 128-B line) then meet their neighbours'
// in ONE L2 instead of leaving three of them as partial lines.  Bijective for any count (the guide's T1 remap).
// Speed / traffic only: robots are independent, any map gives the same results bit for bit.
__host__ __device__ constexpr int xcd_contiguous(int b, int count) {
  return ((b & 7) < (count & 7) ? (b & 7) * ((count >> 3) + 1) : (count & 7) * ((count >> 3) + 1) + ((b & 7) - (count & 7)) * (count >> 3)) + (b >> 3);
}

// ---- robot migration inside a fused launch (SoloConfig::migrate_steps) -----------------------------------------
// A launch of S steps is cut into chunks of c steps; a CHUNK is the unit a wave executes, and between chunks a robot
// is nothing but its 32-real record, its termination counters and two integers in device memory - any wave can
// continue it.  The queue of a launch (one int32 array in device memory, initialised by migration_queue_init before
// the step kernel starts):
//   [r * 32]        head of ring r: the next TICKET (an index into the ring; waves take tickets with one atomic add)
//   [r * 32 + 16]   tail of ring r: the next free ring slot
//   [H + e]         robot e (relative to the launch's first): Gauss-Seidel sweeps so far in this launch (H = kQueueHeader)
//   [H + n ..]      the rings, `q_rings` of them with per_ring = n / q_rings robots x chunks entries each: ring slot ->
//                   robot e ready for its chunk c, as e | c << 24, or -1 = not published yet.  The first per_ring slots
//                   are the ring's robots (chunk 0); every finished chunk but a robot's last publishes ONE further slot,
//                   and every chunk consumes ONE ticket: tickets and slots are both exactly per_ring x chunks, a ring is
//                   a FIFO (robots advance round-robin, so all of them approach the end of the launch together), and the
//                   lowest ticket in flight always finds its slot published (no wave waits for a wave that waits).
// Eight rings, one per XCD (the waves of XCD x serve ring x first: a robot's records and its neighbours' rows of the
// [step][robot][.] arrays stay in one L2), each owning a contiguous eighth of the launch's robots; one ring when the
// robots do not divide evenly.  A wave whose ring has no tickets left goes on to the next ring (so every ring is drained
// whatever the placement of the waves, and the XCDs level out at the end).  (64 rings of 64 robots, measured: slower -
// the pools get too small to balance; the counters' serialisation - read-modify-writes of one address, ~12 ns each -
// is not what a hand-over costs: its chain of device-scope round trips is.)
constexpr int kQueueMaxRings = 8;
constexpr int kQueueHeader = 32 * kQueueMaxRings;
__host__ __device__ constexpr int migration_rings(int n) { return (n % 8 == 0 && n >= 8 * 8) ? 8 : 1; }
__host__ __device__ constexpr int migration_chunks(int steps, int chunk) { return (steps + chunk - 1) / chunk; }
__host__ __device__ constexpr size_t migration_queue_ints(int n, int steps, int chunk) {
  return (size_t)kQueueHeader + (size_t)n + (size_t)n * (size_t)migration_chunks(steps, chunk);
}
// the chunk length a launch really uses: a slot has 7 bits for the chunk index
__host__ __device__ constexpr int migration_chunk_steps(int steps, int chunk) { return migration_chunks(steps, chunk) <= 127 ? chunk : (steps + 126) / 127; }
// entry i of the initialisation (i < migration_queue_ints): order = optional dispatch order of the launch's robots
__host__ __device__ inline void migration_queue_init(int32_t* q, size_t i, int env_base, int n, int rings, int steps, int chunk, const int32_t* order) {
  const int chunks = migration_chunks(steps, chunk), per = n / rings;
  if (i < (size_t)kQueueHeader) {
    const int r = (int)i / 32, w = (int)i % 32;
    q[i] = (r < rings && w == 16) ? per : 0;   // heads 0, tails behind the ring's own robots
  } else if (i < (size_t)kQueueHeader + (size_t)n) {
    q[i] = 0;
  } else {
    const size_t j = i - kQueueHeader - (size_t)n;           // ring slot
    const int r = (int)(j / ((size_t)per * chunks)), k = (int)(j % ((size_t)per * chunks));
    q[i] = k < per ? (order != nullptr ? order[env_base + r * per + k] - env_base : r * per + k) : -1;
  }
}

template <typename T>
struct KBuffers {
  T* state;           // [N][32]
  const T* snapshot;  // [N][32]
  T* targets;         // [N][12]
  const T* actions;   // [N][12] or null
  const T* params;    // [N][4]
  T* traj;            // [count][steps][32] per-step records of a fused launch (robot-major, FIRST ROBOT OF THE LAUNCH first:
                      // a robot's records are contiguous), evaluated by the robot's wave after its last step (lane = step), or null
  T* obs_inline;      // single-step f32 launches (the closed-loop step()): [N][D] / [N] outputs evaluated lane-parallel
  T* reward_inline;   // over the ITEMS of the one step; null otherwise
  // where the outputs of a launch that leaves records go (all optional):
  T* obs_rec;         // observation row of step k, robot e at obs_rec + k * obs_rec_stride + e * D, for k >= obs_from
  T* reward_rec;      // reward of step k, robot e at reward_rec + k * reward_rec_stride + e
  T* view_obs;        // the engine's view [N][D] / [N] / [N]: the LAST step's outputs (a recording rollout's last
  T* view_reward;     // launch, and every rollout that does not record)
  uint8_t* view_done;
  long long obs_rec_stride, reward_rec_stride;
  int32_t obs_from;
  uint8_t* done;      // [N], or [steps][N] with done_stride = N
  int32_t* term_count;  // [N][4]
  double* stats;      // [SOLO_STATS_SHARDS][8]
  const T* terrain;   // [ny][nx] heights, or null = flat plane z = 0
  const int32_t* order;  // [N] workgroup -> robot (solo_engine_set_order), or null = identity
  int32_t* cost;      // [N] Gauss-Seidel sweeps of the robot in this launch, or null
  T* warm;            // [N][64] warm-start cache (SoloConfig::solver_warm_start): every row's impulse at the end of the
                      // robot's previous step, lane layout; null = off
  int32_t num_envs;    // total robots of the engine
  uint32_t flags;
  int32_t env_base;    // first robot of this launch (grid = robots of this launch)
  int32_t count;       // robots of this launch (the XCD-aware workgroup -> robot map needs it: see xcd_contiguous)
  int32_t steps;       // env steps per launch (>= 1)
  // element strides between consecutive steps of a multi-step launch (0: reuse the buffer)
  long long action_stride, done_stride;
  // robot migration (SoloConfig::migrate_steps): the launch's work queue (MigrationQueue below), or null = one wave
  // steps one robot through the whole launch; number of rings (8: one per XCD, or 1); steps per chunk
  int32_t* queue;
  int32_t q_rings, q_chunk;
  int32_t* fault;      // pinned HOST word (device-visible): set by a wave that gives up waiting for its robot (SOLO_ERR_INCOMPLETE)
#ifdef SOLO_STAMPS
  int32_t stamp_row;           // (set by the kernel: the robot this wave steps)
  unsigned long long* stamps;  // [N][32] s_memtime stamps, DIAGNOSTIC builds only (make stamps):
                               // [0..15] absolute stamps of the launch's last step, [16+i] = ticks
                               // spent before stamp i summed over the launch's steps
  unsigned long long* acc;     // the launch's accumulators in LDS (set by the kernel)
#endif
};

// ---- host side: SoloConfig + SoloModel -> KParams ---------------------------------------
inline int validate_model(const SoloModel& m, std::string* err) {
  auto fail = [&](const char* s) { if (err) *err = s; return (int)SOLO_ERR_UNSUPPORTED_MODEL; };
  for (int leg = 0; leg < 4; ++leg) {
    if (m.parent[2 * leg] != 0) return fail("HFE link must hang off the base");
    if (m.parent[2 * leg + 1] != 1 + 2 * leg) return fail("KFE link must hang off its HFE link");
  }
  for (int j = 0; j < SOLO_NUM_DOF; ++j)
    if (m.joint_axis[j][0] != 0.0 || m.joint_axis[j][1] != 1.0 || m.joint_axis[j][2] != 0.0)
      return fail("the HIP kernel is specialised to +y joint axes (Solo8)");
  for (int a = 0; a < 3; ++a)
    if (m.com[0][a] != 0.0) return fail("base frame must be the base CoM frame");
  if (m.num_spheres != SOLO_MAX_SPHERES) return fail("expected 16 collision spheres");
  for (int j = 0; j < SOLO_NUM_DOF; ++j)
    if (!(m.joint_lower[j] < m.joint_upper[j])) return fail("joint limits need lower < upper");
  for (int leg = 0; leg < 4; ++leg) {
    for (int s : {4 * leg, 4 * leg + 1})
      if (m.sphere_body[s] != 1 + 2 * leg && m.sphere_body[s] != 2 + 2 * leg)
        return fail("spheres 4l and 4l+1 must be attached to leg l");
    for (int s : {4 * leg + 2, 4 * leg + 3})
      if (m.sphere_body[s] != 0) return fail("spheres 4l+2 and 4l+3 must be attached to the base");
  }
  return SOLO_OK;
}

template <typename T>
inline void pack_params(const SoloConfig& c, const SoloModel& m, KParams<T>* k) {
  std::memset(k, 0, sizeof(*k));
  k->c.dt = (T)c.dt;
  k->c.inv_dt = (T)(1.0 / c.dt);
  for (int a = 0; a < 3; ++a) k->c.gravity[a] = (T)c.gravity[a];
  k->c.kp_over_dt = (T)(c.motor_kp / c.dt);
  k->c.one_minus_kd = (T)(1.0 - c.motor_kd);
  k->c.motor_impulse = (T)(c.motor_torque_limit * c.dt);
  k->c.lin_damp = (T)c.linear_damping;
  k->c.ang_damp = (T)c.angular_damping;
  k->c.erp_over_dt = (T)(c.contact_erp / c.dt);
  k->c.margin = (T)c.contact_margin;
  k->c.limit_margin = (T)c.joint_limit_margin;
  k->c.action_scale = (T)c.action_scale;
  k->mu_base = (T)c.base_lateral_friction;
  k->c.iterations = c.solver_iterations;
  k->c.auto_reset = c.auto_reset;
  k->c.ulp_tol = c.solver_ulp_tolerance;
  k->c.resid_thr = (T)c.solver_residual_threshold;
  k->c.warm_factor = (T)c.solver_warm_start;
  for (int j = 0; j < SOLO_NUM_JOINTS; ++j) k->c.settle_tgt[j] = (T)c.settle_targets[j];
  k->c.base_mass = (T)m.mass[0];
  for (int a = 0; a < 6; ++a) k->c.base_I[a] = (T)m.inertia[0][a];
  for (int leg = 0; leg < 4; ++leg) {
    LegConst<T>& L = k->leg[leg];
    const int ju = 2 * leg, jl = 2 * leg + 1, bu = 1 + ju, bl = 1 + jl;
    for (int a = 0; a < 3; ++a) {
      L.hip[a] = (T)m.joint_origin[ju][a];
      L.knee[a] = (T)m.joint_origin[jl][a];
      L.link[0][1 + a] = (T)m.com[bu][a];
      L.link[1][1 + a] = (T)m.com[bl][a];
    }
    L.link[0][0] = (T)m.mass[bu];
    L.link[1][0] = (T)m.mass[bl];
    for (int a = 0; a < 6; ++a) { L.link[0][4 + a] = (T)m.inertia[bu][a]; L.link[1][4 + a] = (T)m.inertia[bl][a]; }
    L.limit[0][0] = (T)m.joint_lower[ju]; L.limit[0][1] = (T)m.joint_upper[ju];
    L.limit[1][0] = (T)m.joint_lower[jl]; L.limit[1][1] = (T)m.joint_upper[jl];
  }
  for (int lane = 0; lane < 64; ++lane) k->row[lane].type = ROW_IDLE;
  for (int d = 0; d < SOLO_NUM_DOF; ++d) {
    RowConst<T>& r = k->row[motor_lane(d)];
    r.type = ROW_MOTOR;
    r.dof = d;
  }
  for (int d = 0; d < SOLO_NUM_DOF; ++d) {
    RowConst<T>& r = k->row[16 * (d >> 1) + 14 + (d & 1)];
    r.type = ROW_LIMIT;
    r.dof = d;
  }
  for (int s = 0; s < SOLO_MAX_SPHERES; ++s)
    for (int q = 0; q < 3; ++q) {
      RowConst<T>& r = k->row[sphere_lane(s) + q];
      r.type = ROW_NORMAL + q;
      r.dof = s;
      const int b = m.sphere_body[s];
      r.body = b == 0 ? BODY_BASE : ((b & 1) ? BODY_UPPER : BODY_LOWER);
      for (int a = 0; a < 3; ++a) r.center[a] = (T)m.sphere_center[s][a];
      r.radius = (T)m.sphere_radius[s];
    }
}

template <typename T>
inline int pack_program(const SoloProgram& p, KParams<T>* k, std::string* err) {
  auto fail = [&](const char* s) { if (err) *err = s; return (int)SOLO_ERR_INVALID_ARG; };
  if (p.num_obs < 0 || p.num_obs > SOLO_MAX_OBS) return fail("num_obs out of range");
  if (p.num_reward_ops < 0 || p.num_reward_ops > SOLO_MAX_REWARD_OPS) return fail("num_reward_ops out of range");
  if (p.num_terms < 0 || p.num_terms > SOLO_MAX_TERMS) return fail("num_terms out of range");
  // the reward program must be a well-formed postfix expression leaving one value
  int depth = 0;
  int stack[SOLO_MAX_REWARD_OPS];  // instruction index that produced each stack entry
  int srcs[SOLO_MAX_REWARD_OPS];
  for (int i = 0; i < p.num_reward_ops; ++i) {
    const int op = p.reward[i].op;
    srcs[i] = 0;
    if (op < SOLO_R_CONST || op > SOLO_R_MUL) return fail("bad reward opcode");
    if (op <= SOLO_R_SMALL_CONTROL) stack[depth++] = i;
    else if (op == SOLO_R_SCALE) {
      if (depth < 1) return fail("reward stack underflow");
      srcs[i] = stack[depth - 1];
      stack[depth - 1] = i;
    } else {
      if (depth < 2) return fail("reward stack underflow");
      srcs[i] = stack[depth - 2] | (stack[depth - 1] << 8);
      --depth;
      stack[depth - 1] = i;
    }
    if (depth > 8) return fail("reward stack deeper than 8");
    if (op >= SOLO_R_FLAT_TORSO && op <= SOLO_R_SMALL_CONTROL) {
      // rewards.py:405-417: lower <= upper, margin >= 0
      const double margin = op == SOLO_R_FLAT_TORSO ? p.reward[i].b
                            : (op == SOLO_R_SMALL_CONTROL ? p.reward[i].a : p.reward[i].c);
      if (margin < 0) return fail("Margin must be non-negative");
      const double half = op == SOLO_R_FLAT_TORSO ? p.reward[i].a
                          : (op == SOLO_R_SMALL_CONTROL ? 0.0 : p.reward[i].b);
      if (half < 0) return fail("Lower bound is greater than upper bound");
    }
  }
  if (p.num_reward_ops > 0 && depth != 1) return fail("reward program must leave exactly one value");
  for (int i = 0; i < p.num_obs; ++i)
    if (p.obs[i].src < 0 || p.obs[i].src >= SOLO_SRC_COUNT) return fail("obs source out of range");
  for (int i = 0; i < p.num_terms; ++i)
    if (p.term_kind[i] < SOLO_T_PERPETUAL || p.term_kind[i] > SOLO_T_CONST) return fail("bad termination kind");
  k->c.num_obs = p.num_obs;
  k->c.num_reward_ops = p.num_reward_ops;
  k->c.num_terms = p.num_terms;
  for (int i = 0; i < p.num_obs; ++i) {
    ObsElemK<T>& o = k->obs[i];
    o.src = p.obs[i].src;
    o.flags = p.obs[i].flags;
    o.scale = (T)p.obs[i].scale;
    o.lo = (T)p.obs[i].lo;
    o.hi = (T)p.obs[i].hi;
    const double range = (double)p.obs[i].nhi - (double)p.obs[i].nlo;
    o.nscale = (T)(range != 0.0 ? 2.0 / range : 0.0);
    o.noff = (T)(range != 0.0 ? -2.0 * (double)p.obs[i].nlo / range - 1.0 : 0.0);
  }
  for (int i = 0; i < p.num_reward_ops; ++i) {
    k->reward[i].op = p.reward[i].op;
    k->reward[i].src = srcs[i];
    k->reward[i].a = (T)p.reward[i].a;
    k->reward[i].b = (T)p.reward[i].b;
    k->reward[i].c = (T)p.reward[i].c;
    double margin = 0.0;
    if (p.reward[i].op == SOLO_R_FLAT_TORSO) margin = p.reward[i].b;
    else if (p.reward[i].op == SOLO_R_TORSO_HEIGHT || p.reward[i].op == SOLO_R_HORIZ_SPEED) margin = p.reward[i].c;
    else if (p.reward[i].op == SOLO_R_SMALL_CONTROL) margin = p.reward[i].a;
    k->reward[i].d = (T)(margin != 0.0 ? std::sqrt(-2.0 * std::log(0.1)) / margin : 0.0);
  }
  for (int i = 0; i < SOLO_MAX_TERMS; ++i) {
    k->c.term_kind[i] = p.term_kind[i];
    k->c.term_param[i] = p.term_param[i];
  }
  return SOLO_OK;
}

}  // namespace solo
