// solo_step_kernel.h — the fused Solo8 env step for gfx950: ONE WAVEFRONT PER ROBOT.
//
// Replaces, for N robots at once, the reference's per-step call sequence
//   setJointMotorControlArray + stepSimulation      gym_solo/envs/solo8v2vanilla.py:87-91
//   ObservationFactory.get_obs                      gym_solo/core/obs.py:130-159
//   RewardFactory.get_reward                        gym_solo/core/rewards.py:104-118
//   TerminationFactory.is_terminated                gym_solo/core/termination.py:38-50
//
// Design (written for CDNA4, not translated from anything):
//  * grid = N workgroups of 64 threads; the robot's 128-B state record is ONE coalesced load.
//  * lane = 16*leg + k.  The four 16-lane groups run the leg-local part of the articulated
//    dynamics (kinematics, composite inertias, Newton-Euler bias) in parallel; the base-level
//    sums over the legs go through LDS (27 terms at once) or DPP row broadcasts.
//  * The floating base couples the legs only through a 6x6 block, so the joint-space inverse
//    inertia is applied in factored form: Cholesky of the four 2x2 leg blocks P_l and of the
//    6x6 base Schur complement S = M_bb - sum_l M_bl P_l^-1 M_lb (the base's articulated-body
//    inertia).  Every constraint row r (8 motor rows, 8 joint-limit rows, 3 per touching sphere) is
//    owned by ONE LANE, which whitens its Jacobian against those factors (ghat_r in R^6, hhat_r in
//    R^2), so that the Delassus matrix is  A_sr = ghat_s.ghat_r + [same leg] hhat_s.hhat_r.
//  * The scaled columns of that matrix for the rows that can move are built once per step into register
//    tuples (ColumnBank, solo_wave_ops.h: 64 VGPRs in f32, 128 in f64).  Projected Gauss-Seidel keeps, per lane, the row's candidate, impulse and
//    bounds; the rows that still move are found for all lanes at once (clamp, subtract, compare:
//    the compare's lane mask is the set) and only those are visited, in solver order (non-contact
//    rows, normal rows, friction rows), each with one register-indexed move for its column (see
//    physics_solve; the loop itself is written in assembly for both precisions: solo_pgs_gfx950.h - a serial
//    chain issued by one wave at ~6 cycles per instruction, so its time is its instruction count).  All
//    branching is wave-uniform: the whole wave belongs to one robot.
//  * The ~50 scalars a step reads, the per-leg and per-row tables live in LDS for the launch; the
//    rarely used kernel arguments are re-read from the kernarg segment where they are used.
//  * Termination (TimeBased counters) and the auto-reset belong to the step; observations, rewards and
//    episodic returns do NOT: a fused launch leaves one 128-B record per robot-step and the robot's wave
//    evaluates them AFTER its last step, 32 steps at a time with lane = step (solo_outputs.h's per-item
//    functions; round 3 - rounds 1-2 ran separate output kernels, one thread per robot-step, after the
//    launch: ~14 % of a 20-step rollout).  Only a single-step f32 launch - the closed-loop step() -
//    evaluates its outputs in place, lane = item, with the same functions.
// No MFMA: there is no dense contraction here (14 dofs, <= 64 rows per robot).
//
// The including translation unit must provide solo::lane_id/block_id/wave_sync/wave_readlane/
// wave_sum_group16/wave_sum_all/wave_reduce_rows/wave_lane_below/wave_slot_below/wave_count_below/wave_push/wave_pull/wave_ballot/wave_cold_args/RowDot<T>/
// ColumnBank<T>/Real<T>/stats_add (solo_wave_ops.h on the GPU, tests/emu/wave_emu.h on the CPU emulator).
#pragma once

#include "solo_kernel_params.h"
#include "solo_outputs.h"

// In-kernel phase stamps exist only in the diagnostic build (make -C gym_solo_amd/csrc stamps);
// in the product build the macro expands to nothing.
#if defined(SOLO_STAMPS) && defined(SOLO_STAMPS_LIGHT)
// light variant (make stamps_light): only the launch's first and last stamp, nothing per step - on the 100-MHz
// real-time counter, which all XCDs share (s_memtime counts per XCD: tools/gpu_critical_path.py compares across them)
#define SOLO_STAMP(B, i)                                                                          \
  do {                                                                                            \
    if (((i) == 0 || (i) == 14) && solo::lane_id() == 0)                                          \
      (B).stamps[(size_t)(B).stamp_row * 32 + (i)] = __builtin_amdgcn_s_memrealtime();         \
  } while (0)
#elif defined(SOLO_STAMPS)
// (make stamps_epilogue, -DSOLO_STAMPS_EPILOGUE: stamps 1 .. 12 sit INSIDE the output epilogue instead of the step)
#ifdef SOLO_STAMPS_EPILOGUE
#define SOLO_STAMP_ON(i) ((i) == 0 || (i) >= 13)
#define SOLO_STAMP_E(B, i) SOLO_STAMP_AT(B, i)
#else
#define SOLO_STAMP_ON(i) true
#endif
#define SOLO_STAMP(B, i) do { if (SOLO_STAMP_ON(i)) SOLO_STAMP_AT(B, i); } while (0)
#define SOLO_STAMP_AT(B, i)                                                                       \
  do {                                                                                            \
    if (solo::lane_id() == 0) {                                                                   \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                 \
      (B).stamps[(size_t)(B).stamp_row * 32 + (i)] = t_;                                            \
      (B).acc[(i)] += t_ - (B).acc[16];                                                           \
      (B).acc[16] = t_;                                                                           \
    }                                                                                             \
  } while (0)
#else
#define SOLO_STAMP(B, i) do {} while (0)
#endif
#ifndef SOLO_STAMP_E
#define SOLO_STAMP_E(B, i) do {} while (0)
#endif

// test hook (CPU emulator builds only): called once per Gauss-Seidel sweep
#ifndef SOLO_PGS_SWEEP_HOOK
#define SOLO_PGS_SWEEP_HOOK(it, pend, lam, v) do {} while (0)
#endif

namespace solo {

template <typename T> struct V3 { T x, y, z; };
template <typename T> __device__ __forceinline__ V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> __device__ __forceinline__ V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> __device__ __forceinline__ V3<T> operator*(T s, V3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename T> __device__ __forceinline__ T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> __device__ __forceinline__ V3<T> cross(V3<T> a, V3<T> b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// rotation about +y by the angle whose cos/sin are (c, s): child -> parent coordinates
template <typename T> __device__ __forceinline__ V3<T> roty(T c, T s, V3<T> v) {
  return {c * v.x + s * v.z, v.y, c * v.z - s * v.x};
}
// y_hat x r
template <typename T> __device__ __forceinline__ V3<T> ycross(V3<T> r) { return {r.z, T(0), -r.x}; }
// symmetric 3x3 (xx yy zz xy xz yz) times vector
template <typename T> __device__ __forceinline__ V3<T> symmul(const T* I, V3<T> v) {
  return {I[0] * v.x + I[3] * v.y + I[4] * v.z, I[3] * v.x + I[1] * v.y + I[5] * v.z,
          I[4] * v.x + I[5] * v.y + I[2] * v.z};
}
// R I R^T for R = Ry
template <typename T> __device__ __forceinline__ void rot_inertia_y(T c, T s, const T* I, T* o) {
  const T cc = c * c, ss = s * s, cs = c * s;
  o[0] = cc * I[0] + T(2) * cs * I[4] + ss * I[2];
  o[1] = I[1];
  o[2] = ss * I[0] - T(2) * cs * I[4] + cc * I[2];
  o[3] = c * I[3] + s * I[5];
  o[4] = cs * (I[2] - I[0]) + (cc - ss) * I[4];
  o[5] = c * I[5] - s * I[3];
}
template <typename T> __device__ __forceinline__ T sum_over_group16(T x) { return wave_sum_group16(x); }
template <typename T> __device__ __forceinline__ V3<T> select(bool p, V3<T> a, V3<T> b) { return {p ? a.x : b.x, p ? a.y : b.y, p ? a.z : b.z}; }
// the two halves (k < 8 / k >= 8) of a 16-lane row: sum of both, or the lower half's value in both
template <typename T> __device__ __forceinline__ T both_halves(T x) { return x + wave_other_half16(x); }
template <typename T> __device__ __forceinline__ V3<T> both_halves(V3<T> v) { return {both_halves(v.x), both_halves(v.y), both_halves(v.z)}; }
template <typename T> __device__ __forceinline__ T from_lower(T x) { return wave_from_lower_half16(x); }
template <typename T> __device__ __forceinline__ V3<T> from_lower(V3<T> v) { return {from_lower(v.x), from_lower(v.y), from_lower(v.z)}; }


}  // namespace solo
namespace solo {
// sweeps per step above which a robot's wave is pinned to the top issue priority (physics_solve).  f32: 8 (thresholds
// 4 / 8 / 20 / 40 measured in round 2: 1.063 / 1.060 / 1.027 / 1.021e8).  f64 runs to its fixed point in more sweeps
// (mean 9.4 per robot-step against 4.9): with 8 nearly half of the robots were "slow"; 12 is worth +2 % on 250-step
// launches, nothing at 20 steps or one (8 / 12 / 16 / 24 / 40: 1.68 / 1.72 / 1.70 / 1.69 / 1.68e8, profiles/round4_ab.log)
#ifndef SOLO_PRIO_SWEEPS_F64
#define SOLO_PRIO_SWEEPS_F64 12
#endif
#ifndef SOLO_QUEUE_SPINS
#define SOLO_QUEUE_SPINS (1 << 22)   // polls of a ring slot before a wave gives up (SOLO_ERR_INCOMPLETE); the CPU emulator's fault-injection test builds with fewer
#endif
#ifndef SOLO_PRIO_BY_ROWS
#define SOLO_PRIO_BY_ROWS 1   // (0: the A/B build)
#endif
#ifndef SOLO_PRIO_ROWS_1
#define SOLO_PRIO_ROWS_1 11
#define SOLO_PRIO_ROWS_2 17
#define SOLO_PRIO_ROWS_3 23
#endif
__device__ __forceinline__ int priority_by_rows(int rows) { return rows > SOLO_PRIO_ROWS_3 ? 3 : (rows > SOLO_PRIO_ROWS_2 ? 2 : (rows > SOLO_PRIO_ROWS_1 ? 1 : 0)); }
template <typename T> constexpr int kPrioSweeps = sizeof(T) == 4 ? 8 : SOLO_PRIO_SWEEPS_F64;
// The block of LDS that holds a step's constraint rows - f32: [64 rows][ghat 6, hhat 2] and the joint-space parts again by
// leg slot [64][4 legs x 2]; f64 (slot space): [64 slots][ghat 6, hhat 2], the legs are tags of their own - and is the
// scratch of whatever runs while the rows are dead: the f64 post-solve reduction (kReduceScratch) and the output
// epilogue's reward values [SOLO_MAX_REWARD_OPS][steps of a pass].  f64: 896 reals = 28 steps per pass (at FOUR waves
// per SIMD - the A/B build - what fits 10 KB of LDS: 25).
#ifndef SOLO_F64_WAVES
#define SOLO_F64_WAVES 4   // (round 5: FOUR waves per SIMD - 128 VGPRs, 10240 B of LDS; -DSOLO_F64_WAVES=3 / 2: the A/B builds, make w3 / w2)
#endif
template <typename T> constexpr int kRowBlockReals = sizeof(T) == 4 ? 64 * 8 + 64 * 8 : (SOLO_F64_WAVES >= 4 ? 800 : 896);
#ifndef SOLO_W4_PIPELINED_BUILD
#define SOLO_W4_PIPELINED_BUILD 0
#endif
#ifndef SOLO_W4_HYBRID_BUILD
#define SOLO_W4_HYBRID_BUILD 1   // (0: the A/B build - no column is pipelined at four waves per SIMD, as in round 5)
#endif
constexpr int kLegSlots = 26;  // per-leg parking lot in LDS (see physics_solve): 0-11 K, 12-14 Lp factors, 15-16 unconstrained joint rates, 17-18 q, 19-22 cos / sin of the two link angles, 23-24 P^-1 h

// one lane's constraint-row constants, as the step reads them from LDS (staged from KParams::row once per launch)
template <typename T> struct RowView {
  int type, body;   // RowType, BodyKind
  const T* geo;     // sphere centre [3] in its body frame, radius
};
// the per-launch tables in LDS (wave-uniform addresses): what a lane reads of them is a function of its lane number, and the
// f64 step re-derives that at the head of the row phase instead of carrying three addresses across the dynamics phase
template <typename T> struct StepTables {
  const LegConst<T>* legc;    // [4]
  const int32_t* rowtype;     // [64]: RowConst::type | RowConst::body << 4 | geometry entry << 8 | leg_sum_entry(lane) << 16
  const T (*rowgeo)[4];       // [SOLO_MAX_SPHERES + 1]
  __device__ __forceinline__ RowView<T> row(int lane) const {
    const int tb = rowtype[lane];
    return {tb & 15, (tb >> 4) & 15, rowgeo[(tb >> 8) & 255]};
  }
};

// rows of one sweep, in solver order ([recalled] btMultiBodyConstraintSolver::solveSingleIteration):
// the non-contact rows (joint motors, joint limits), then ALL normal contact rows, then ALL friction rows -
// as lane masks of the fixed lane = row layout (solo_kernel_params.h); the slot-space solver builds its own per step
constexpr unsigned long long kPhaseLanes[3] = {0xc003c003c003c003ull,    // motors k = 0, 1 and joint limits k = 14, 15, leg by leg
                                               0x0924092409240924ull,    // normals  k = 2, 5, 8, 11
                                               0x36d836d836d836d8ull};   // friction k = 3, 4, 6, 7, 9, 10, 12, 13

// The Gauss-Seidel iteration, C++ DEFINITION (see physics_solve for what the per-lane state means).  The CPU emulator
// and the -DSOLO_PGS_NO_ASM test build run it; on the GPU the product runs solo_pgs_gfx950.h's assembly - same rows,
// same order, same arithmetic, compared bit for bit by tests/test_gpu_pgs_asm.py - and this function only where the
// assembly does not go: kFromLds, the OVERFLOW path of the slot-space solver (more live rows than the column bank has
// slots: every column is evaluated from LDS when it is used - ColumnBank::column, the expression build() stores).
// Returns the number of sweeps.
template <typename T, bool kResid, bool kFromLds>
__device__ __forceinline__ int pgs_solve_cpp(const ColumnBank<T>& A, T& v, T& lamv, T& cand, T& dl, unsigned long long& pend, T& lo, T& hi,
                                             T tol_rel, int lane, T mu, bool is_tan1, bool is_tangent, unsigned long long ph0,
                                             unsigned long long ph1, unsigned long long ph2, int iters, T diag, T resid_thr, int& n_changed) {
  using R = Real<T>;
  int it = 0;
#pragma unroll 1
  for (; it < iters && pend != 0ull; ++it) {  // nothing pending at the start of a sweep: converged
    // (the register banks of the matrix are walked one after the other - static bank per loop - which
    // keeps the rows of a phase in ascending lane order)
    const T lam_sweep_start = lamv;
    bool normals_moved = false;  // (wave-uniform)
#pragma unroll
    for (int phase = 0; phase < 3; ++phase) {
      const unsigned long long phase_lanes = phase == 0 ? ph0 : (phase == 1 ? ph1 : ph2);
      if (phase == 2 && normals_moved) {
        // the friction rows of a contact are limited by mu x the normal impulse it holds NOW: the
        // normal rows are done for this sweep, so all limits are refreshed at once (a friction
        // row's normal row sits one or two lanes below it: DPP shifts) instead of per moved row
        // - and only in a sweep that moved a normal row (all limits start at mu x 0 = 0)
        T n1, n2;
        if constexpr (ColumnBank<T>::kCompact) { n1 = wave_slot_below<1>(lamv); n2 = wave_slot_below<2>(lamv); }
        else { n1 = wave_lane_below<1>(lamv); n2 = wave_lane_below<2>(lamv); }
        const T lim = mu * (is_tan1 ? n1 : n2);
        lo = is_tangent ? -lim : lo;
        hi = is_tangent ? lim : hi;
        cand = R::clamp(v, lo, hi);
        dl = cand - lamv;
        pend = wave_ballot(R::abs(dl) > R::abs(lamv) * tol_rel);
      }
      if ((pend & phase_lanes) == 0ull) continue;  // nothing of this phase moves: one test for its banks
      if (phase == 1) normals_moved = true;
#pragma unroll
      for (int bank = 0; bank < (kFromLds ? 1 : ColumnBank<T>::kBanks); ++bank) {
        unsigned long long window = kFromLds ? phase_lanes : (phase_lanes & ColumnBank<T>::bank_lanes(bank));
#pragma unroll 1
        while ((pend & window) != 0ull) {
          const int r = __builtin_ctzll(pend & window);  // wave-uniform: the row to update
          window &= ~((2ull << r) - 1ull);               // the cursor moves past it
          const T col = kFromLds ? A.column(r) : A.get(bank, r);  // column r of the scaled matrix (0 for the row itself)
          const T delta = wave_readlane(dl, r);
          v = R::fma(col, delta, v);
          lamv = (lane == r) ? cand : lamv;
          cand = R::clamp(v, lo, hi);
          dl = cand - lamv;
          pend = wave_ballot(R::abs(dl) > R::abs(lamv) * tol_rel);
          ++n_changed;
        }
      }
    }
    SOLO_PGS_SWEEP_HOOK(it, pend, lamv, v);
    const T dvel = (lamv - lam_sweep_start) * diag;
    if (kResid && wave_ballot(dvel * dvel > resid_thr) == 0ull) { ++it; break; }  // pybullet's residual threshold (see physics_solve)
  }
  return it;
}

// ------------------------------------------------------------------------------------------
// physics: A3 + A4 of SURVEY.md §8a.  physics_solve reads s_state (old) and returns this lane's
// constraint impulse; physics_finish writes s_state (new).
// ------------------------------------------------------------------------------------------
template <typename T, bool kResid, bool kPipelinedBuild, typename FetchTarget>
__device__ __forceinline__ T physics_solve(const StepConst<T>& C, const KBuffers<T>& B, const StepTables<T>& tabs,
                                           const T* s_state, T my_target, FetchTarget&& fetch_target, T* s_rowvec, T (*s_hext)[8], unsigned char* s_rowleg,
                                           T* s_keep, T (*s_leg)[kLegSlots], const T* s_math, T mu, T mass_scale, int lane, int& row_at, bool& target_bad,
                                           int& prio_sweeps, int& prio_steps, int& prio_rot, T warm_in = T(0), bool warm_on = false) {
  constexpr bool kCompact = ColumnBank<T>::kCompact;   // the solver runs in slot space (see "slot space" below)
  constexpr int kRS = ColumnBank<T>::kRowStride;       // reals per row vector in s_rowvec
  using R = Real<T>;
  int leg = lane >> 4, k = lane & 15;
  const LegConst<T>* Lp = &tabs.legc[leg];
#define L (*Lp)

  // ---- leg-local kinematics.  Each 16-lane row works on its own leg, and the two HALVES of the
  //      row on the leg's two links: lanes k < 8 carry the upper link, k >= 8 the lower link
  //      (+ welded foot) through the same instructions; per-leg quantities are the sum of the two
  //      halves (x + the other half's x, one DPP add: identical bits in both halves).
  const bool lower = (k & 8) != 0;
  const T bm = lower ? T(1) : T(0);
  const T q1 = s_state[SOLO_S_Q + 2 * leg], q2 = s_state[SOLO_S_Q + 2 * leg + 1];
  const T qd1 = s_state[SOLO_S_QD + 2 * leg], qd2 = s_state[SOLO_S_QD + 2 * leg + 1];
  // ONE sincos per lane - of its own link's absolute angle (q1 on the upper half, q1 + q2 on the lower) -
  // and each link's pair broadcast over both halves of the 16-lane row (bank-masked DPP moves)
  T sinb, cosb;  // this half's link orientation
  R::sincos(lower ? q1 + q2 : q1, &sinb, &cosb, s_math);
  const T s1 = wave_from_upper_half16(sinb), c1 = wave_from_upper_half16(cosb);
  const T s12 = wave_from_lower_half16(sinb), c12 = wave_from_lower_half16(cosb);
  const V3<T> o1 = {L.hip[0], L.hip[1], L.hip[2]};
  const V3<T> o2 = o1 + roty(c1, s1, V3<T>{L.knee[0], L.knee[1], L.knee[2]});
  const V3<T> ob = select(lower, o2, o1);                // ... and joint origin
  const T* body = L.link[lower ? 1 : 0];                 // m, com[3], inertia[6] of this half's link
  const T mB = body[0];
  const V3<T> c = ob + roty(cosb, sinb, V3<T>{body[1], body[2], body[3]});
  T I[6];
  rot_inertia_y(cosb, sinb, body + 4, I);

  SOLO_STAMP(B, 2);
  // ---- joint-space inertia blocks of the leg (composite-rigid-body, closed form) -----------
  const V3<T> r1 = c - o1, r2 = c - o2;  // r2 (and everything about joint 2) is meaningful on the lower half
  const V3<T> t1 = ycross(r1), t2 = ycross(r2);
  const V3<T> Iy = {I[3], I[1], I[5]};  // I * y_hat
  // column of M for dof 1 / dof 2: [n; f] = [angular momentum about the base origin; linear]
  const V3<T> f1 = both_halves(mB * t1);
  const V3<T> n1 = both_halves(Iy + mB * cross(c, t1));
  const T P11 = both_halves(mB * dot(t1, t1) + I[1]);
  const V3<T> f2 = from_lower(mB * t2);
  const V3<T> n2 = from_lower(Iy + mB * cross(c, t2));
  const T P12 = from_lower(mB * dot(t1, t2) + I[1]);
  const T P22 = from_lower(mB * dot(t2, t2) + I[1]);
  // Cholesky of the 2x2 leg block, W = Lp^-1 [F1;F2], K = P^-1 M_lb = Lp^-T W
  const T iL11 = R::rsqrt(P11);
  const T L21 = P12 * iL11;
  const T iL22 = R::rsqrt(P22 - L21 * L21);
  T W1[6] = {n1.x * iL11, n1.y * iL11, n1.z * iL11, f1.x * iL11, f1.y * iL11, f1.z * iL11};
  const T F2[6] = {n2.x, n2.y, n2.z, f2.x, f2.y, f2.z};
  T W2[6], K1[6], K2[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    W2[i] = (F2[i] - L21 * W1[i]) * iL22;
    K2[i] = W2[i] * iL22;
    K1[i] = (W1[i] - L21 * K2[i]) * iL11;
  }
  // f64: PARK EARLY.  The register peak of the whole step is the 6x6 base factorisation below with everything the legs
  // have produced still live (221 VGPRs when the compiler may have them; 168 = three waves per SIMD is the budget):
  // what the legs computed for later phases goes to its place in LDS NOW - the two wave_syncs of the leg sum below lie
  // between these stores and the loads that bring it back (so the compiler cannot forward the registers) - and the
  // row phase re-derives the base rotation from the quaternion instead of keeping nine values across the factorisation.
  // (f32 fits its 128 VGPRs without: it keeps the values and parks once, at the end of the dynamics phase.)
  constexpr bool kPark = sizeof(T) == 8;
  if constexpr (kPark) {
    if (k == 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) { s_leg[leg][i] = K1[i]; s_leg[leg][6 + i] = K2[i]; }
      s_leg[leg][12] = iL11; s_leg[leg][13] = L21; s_leg[leg][14] = iL22;
      s_leg[leg][17] = q1; s_leg[leg][18] = q2;
      s_leg[leg][19] = c1; s_leg[leg][20] = s1; s_leg[leg][21] = c12; s_leg[leg][22] = s12;
    }
  }
  // leg composite about the base origin
  const T mleg = L.link[0][0] + L.link[1][0];
  const V3<T> mc = both_halves(mB * c);
  T IO[6];
  IO[0] = both_halves(I[0] + mB * (c.y * c.y + c.z * c.z));
  IO[1] = both_halves(I[1] + mB * (c.x * c.x + c.z * c.z));
  IO[2] = both_halves(I[2] + mB * (c.x * c.x + c.y * c.y));
  IO[3] = both_halves(I[3] - mB * c.x * c.y);
  IO[4] = both_halves(I[4] - mB * c.x * c.z);
  IO[5] = both_halves(I[5] - mB * c.y * c.z);
  if constexpr (kPark) {
    // ... and what the base-level sum below is made of (the row-vector block is dead here: it is the scratch of that sum)
    if (k == 0) {
      T* mine = s_rowvec + leg * 32;
#pragma unroll
      for (int i = 0; i < 6; ++i) { mine[i] = IO[i]; mine[17 + i] = W1[i]; mine[24 + i] = W2[i]; }
      mine[6] = mc.x; mine[7] = mc.y; mine[8] = mc.z; mine[9] = mleg; mine[10] = T(0);
    }
    // ... and the motor lanes' targets are fetched from global memory HERE: behind the register peak, thousands of cycles
    // in front of the motor rows that consume them
    my_target = fetch_target(wave_fresh_lane());
  }

  SOLO_STAMP(B, 3);
  // ---- base: rotation (body -> world), velocities and gravity in base coordinates ----------
  const T qx = s_state[SOLO_S_QUAT], qy = s_state[SOLO_S_QUAT + 1], qz = s_state[SOLO_S_QUAT + 2], qw = s_state[SOLO_S_QUAT + 3];
  const T r00 = T(1) - T(2) * (qy * qy + qz * qz), r01 = T(2) * (qx * qy - qw * qz), r02 = T(2) * (qx * qz + qw * qy);
  const T r10 = T(2) * (qx * qy + qw * qz), r11 = T(1) - T(2) * (qx * qx + qz * qz), r12 = T(2) * (qy * qz - qw * qx);
  const T r20 = T(2) * (qx * qz - qw * qy), r21 = T(2) * (qy * qz + qw * qx), r22 = T(1) - T(2) * (qx * qx + qy * qy);
  const V3<T> ww = {s_state[SOLO_S_ANGVEL], s_state[SOLO_S_ANGVEL + 1], s_state[SOLO_S_ANGVEL + 2]};
  const V3<T> vw = {s_state[SOLO_S_LINVEL], s_state[SOLO_S_LINVEL + 1], s_state[SOLO_S_LINVEL + 2]};
  const V3<T> gw = {C.gravity[0], C.gravity[1], C.gravity[2]};
  // R^T v
  const V3<T> om = {r00 * ww.x + r10 * ww.y + r20 * ww.z, r01 * ww.x + r11 * ww.y + r21 * ww.z, r02 * ww.x + r12 * ww.y + r22 * ww.z};
  const V3<T> vb = {r00 * vw.x + r10 * vw.y + r20 * vw.z, r01 * vw.x + r11 * vw.y + r21 * vw.z, r02 * vw.x + r12 * vw.y + r22 * vw.z};
  const V3<T> gb = {r00 * gw.x + r10 * gw.y + r20 * gw.z, r01 * gw.x + r11 * gw.y + r21 * gw.z, r02 * gw.x + r12 * gw.y + r22 * gw.z};
  const V3<T> nb = {r20, r21, r22};  // world z (ground normal) in base coordinates

  // ---- bias forces of the leg: Newton-Euler with classical accelerations in the frame that
  //      coincides with the base at this instant (gravity + Bullet-style damping included) ----
  const V3<T> wU = {om.x, om.y + qd1, om.z};
  const V3<T> wB = {om.x, wU.y + bm * qd2, om.z};                               // this half's link
  const V3<T> aU = {-qd1 * om.z, T(0), qd1 * om.x};                             // om x (qd1 y)
  const V3<T> a = {aU.x - (bm * qd2) * wU.z, T(0), aU.z + (bm * qd2) * wU.x};   // (+ wU x (qd2 y))
  // centripetal terms as w x (w x r) = w (w.r) - |w|^2 r
  const T om2 = dot(om, om), wU2 = dot(wU, wU), wB2 = dot(wB, wB);
  const V3<T> a_o1 = dot(om, o1) * om - om2 * o1;
  const V3<T> d12 = o2 - o1;
  const V3<T> a_o2 = a_o1 + cross(aU, d12) + (dot(wU, d12) * wU - wU2 * d12);
  const V3<T> r = c - ob;
  const V3<T> a_c = select(lower, a_o2, a_o1) + cross(a, r) + (dot(wB, r) * wB - wB2 * r);
  const V3<T> v_c = vb + cross(om, c) + qd1 * t1 + (bm * qd2) * t2;
  const T kl = C.lin_damp, ka = C.ang_damp;
  const T dB = kl * (T(1) + R::sqrt(dot(v_c, v_c)));
  const T eB = ka * (T(1) + R::sqrt(wB2));
  const V3<T> F = mB * (a_c - gb + dB * v_c);
  const V3<T> Iw = symmul(I, wB);
  const V3<T> N = symmul(I, a) + cross(wB, Iw) + eB * Iw;
  // joint torques y . (N + r x F): both links load joint 1, the lower one joint 2
  const T h1 = both_halves(N.y + (r1.z * F.x - r1.x * F.z));
  const T h2 = from_lower(N.y + (r2.z * F.x - r2.x * F.z));
  const V3<T> Fleg = both_halves(F);
  const V3<T> Nleg = both_halves(N + cross(c, F));
  // e = Lp^-1 h ; y = Lp^-T e = P^-1 h
  const T e1 = h1 * iL11, e2 = (h2 - L21 * e1) * iL22;
  const T y2 = e2 * iL22, y1 = (e1 - L21 * y2) * iL11;
  if constexpr (kPark) { if (k == 0) { s_leg[leg][23] = y1; s_leg[leg][24] = y2; } }

  SOLO_STAMP(B, 4);
  // ---- base level: Schur complement S and right-hand side, summed over the four legs -------
  // leg composite about the base origin
  T S[6][6];  // lower triangle used
  S[0][0] = IO[0]; S[1][0] = IO[3]; S[1][1] = IO[1]; S[2][0] = IO[4]; S[2][1] = IO[5]; S[2][2] = IO[2];
  S[3][0] = T(0);  S[3][1] = mc.z;  S[3][2] = -mc.y;
  S[4][0] = -mc.z; S[4][1] = T(0);  S[4][2] = mc.x;
  S[5][0] = mc.y;  S[5][1] = -mc.x; S[5][2] = T(0);
  S[3][3] = mleg; S[4][3] = T(0); S[4][4] = mleg; S[5][3] = T(0); S[5][4] = T(0); S[5][5] = mleg;
  T rhs[6] = {-Nleg.x, -Nleg.y, -Nleg.z, -Fleg.x, -Fleg.y, -Fleg.z};
  // Sum of the 27 per-leg terms over the four legs, through LDS: one lane per leg posts its
  // terms, lane e adds the four copies of term e, everyone reads the totals back as broadcasts
  // (~30 instructions; 27 in-register cross-row sums cost six each).  The row-vector array is
  // not live yet and serves as the scratch: part[4][28], tot[28].
  if constexpr (kPark) {
    // f64 (round 5): the legs post what the terms are MADE of (31 values) and lane e evaluates term e of all four legs
    // (leg_sum_entry, solo_kernel_params.h): 3 fused multiply-adds per leg on 27 lanes instead of 60 on all 64 - and W1,
    // W2, the composite inertia and the bias wrench stop being live in every lane across this phase
    T* part = s_rowvec;           // [4][32]
    T* tot = part + 4 * 32;
    if (k == 0) {   // (the inertia part was posted when it was computed, above)
      T* mine = part + leg * 32;
#pragma unroll
      for (int i = 0; i < 6; ++i) mine[11 + i] = rhs[i];
      mine[23] = -e1; mine[30] = -e2;
    }
    wave_sync();
    // (the lane number - and with it every per-lane LDS address - is derived afresh here: nothing of the kind is live
    // across the leg phase's register peak)
    const int ls = wave_fresh_lane();
    lane = ls; leg = ls >> 4; k = ls & 15;
    if (ls < 27) {
      const int tb = tabs.rowtype[ls] >> 16;
      const int bi = tb & 31, ii = (tb >> 6) & 7, jj = (tb >> 9) & 7;
      T t[4];
#pragma unroll
      for (int l2 = 0; l2 < 4; ++l2) {
        const T* p = part + 32 * l2;
        const T b = (tb & 32) ? -p[bi] : p[bi];
        t[l2] = R::fma(-p[24 + ii], p[24 + jj], R::fma(-p[17 + ii], p[17 + jj], b));
      }
      tot[ls] = (t[0] + t[1]) + (t[2] + t[3]);
    }
    wave_sync();
    int o = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int j = 0; j <= i; ++j) S[i][j] = tot[o++];
      rhs[i] = tot[21 + i];
    }
  } else {
    T* part = s_rowvec;
    T* tot = part + 4 * 28;
    if (k == 0) {
      T* mine = part + leg * 28;
      int o = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) mine[o++] = S[i][j] - W1[i] * W1[j] - W2[i] * W2[j];
        mine[21 + i] = rhs[i] + W1[i] * e1 + W2[i] * e2;
      }
    }
    wave_sync();
    if (lane < 27) tot[lane] = (part[lane] + part[28 + lane]) + (part[56 + lane] + part[84 + lane]);
    wave_sync();
    int o = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int j = 0; j <= i; ++j) S[i][j] = tot[o++];
      rhs[i] = tot[21 + i];
    }
    // (the sync after parking the factors below orders these reads before the row phase's writes)
  }
  // the base body itself (mass / inertia scaled per env for domain randomisation)
  if constexpr (sizeof(T) == 8) mass_scale = s_keep[28];  // (staged by the kernel's prologue; the friction coefficient is fetched at the solver)
  const T dt = C.dt;
  {
    const T m0 = C.base_mass * mass_scale;
    T I0[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) I0[i] = C.base_I[i] * mass_scale;
    S[0][0] += I0[0]; S[1][0] += I0[3]; S[1][1] += I0[1]; S[2][0] += I0[4]; S[2][1] += I0[5]; S[2][2] += I0[2];
    S[3][3] += m0; S[4][4] += m0; S[5][5] += m0;
    const V3<T> Iw = symmul(I0, om);
    const V3<T> N0 = cross(om, Iw) + (ka * (T(1) + R::sqrt(om2))) * Iw;
    const V3<T> F0 = m0 * ((kl * (T(1) + R::sqrt(dot(vb, vb)))) * vb - gb);
    rhs[0] -= N0.x; rhs[1] -= N0.y; rhs[2] -= N0.z;
    rhs[3] -= F0.x; rhs[4] -= F0.y; rhs[5] -= F0.z;
  }
  SOLO_STAMP(B, 5);
  // Cholesky S = C C^T (C lower, stored in S; iC = 1/diag)
  T iC[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    T d = S[j][j];
#pragma unroll
    for (int m = 0; m < j; ++m) d -= S[j][m] * S[j][m];
    iC[j] = R::rsqrt(d);
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      T s = S[i][j];
#pragma unroll
      for (int m = 0; m < j; ++m) s -= S[i][m] * S[j][m];
      S[i][j] = s * iC[j];
    }
  }
  // unconstrained acceleration: x_b = S^-1 rhs ; qdd_l = -y_l - K_l x_b
  T xb[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    T s = rhs[i];
#pragma unroll
    for (int m = 0; m < i; ++m) s -= S[i][m] * xb[m];
    xb[i] = s * iC[i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    T s = xb[i];
#pragma unroll
    for (int m = i + 1; m < 6; ++m) s -= S[m][i] * xb[m];
    xb[i] = s * iC[i];
  }
  T kx1 = T(0), kx2 = T(0);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    if constexpr (kPark) { kx1 += s_leg[leg][i] * xb[i]; kx2 += s_leg[leg][6 + i] * xb[i]; }
    else { kx1 += K1[i] * xb[i]; kx2 += K2[i] * xb[i]; }
  }
  // velocities after the unconstrained update (semi-implicit Euler)
  const T ub[6] = {om.x + dt * xb[0], om.y + dt * xb[1], om.z + dt * xb[2],
             vb.x + dt * xb[3], vb.y + dt * xb[4], vb.z + dt * xb[5]};
  T us1, us2;
  if constexpr (kPark) {
    us1 = s_state[SOLO_S_QD + 2 * leg] + dt * (-s_leg[leg][23] - kx1);
    us2 = s_state[SOLO_S_QD + 2 * leg + 1] + dt * (-s_leg[leg][24] - kx2);
  } else {
    us1 = qd1 + dt * (-y1 - kx1);
    us2 = qd2 + dt * (-y2 - kx2);
  }

  // Park the factors and the unconstrained velocities in LDS NOW: the row phase below and the
  // post-solve phase read them back from there (wave-uniform / per-leg broadcasts), so that
  // ~45 values stop occupying VGPRs while the rows and the Delassus matrix are built and the
  // Gauss-Seidel loop keeps only its Delassus row live (128 VGPRs -> 4 waves/SIMD in f32).
  if (lane == 0) {
    int o = 0;
#pragma unroll
    for (int i = 1; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < i; ++j) s_keep[o++] = S[i][j];
#pragma unroll
    for (int i = 0; i < 6; ++i) { s_keep[15 + i] = iC[i]; s_keep[21 + i] = ub[i]; }
  }
  if (k == 0) {
    if constexpr (!kPark) {
#pragma unroll
      for (int i = 0; i < 6; ++i) { s_leg[leg][i] = K1[i]; s_leg[leg][6 + i] = K2[i]; }
      s_leg[leg][12] = iL11; s_leg[leg][13] = L21; s_leg[leg][14] = iL22;
      s_leg[leg][17] = q1; s_leg[leg][18] = q2;
    }
    s_leg[leg][15] = us1; s_leg[leg][16] = us2;
  }
  wave_sync();

  SOLO_STAMP(B, 6);
  // ---- constraint rows: one per lane --------------------------------------------------------
  // what the row phase reads of the kinematics: kept in registers (f32) or brought back from LDS / re-derived (f64: above)
  T m00 = r00, m01 = r01, m02 = r02, m10 = r10, m11 = r11, m12 = r12;
  V3<T> nbr = nb, p1 = o1, p2 = o2;
  T ck1 = c1, sk1 = s1, ck12 = c12, sk12 = s12, qa1 = q1, qa2 = q2;
  if constexpr (kPark) {
    // (f64: the lane number - and with it every per-lane table address - is derived afresh: nothing of the kind is live
    // across the dynamics phase)
    lane = wave_fresh_lane(); leg = lane >> 4; k = lane & 15;
    Lp = &tabs.legc[leg];
    const T ux = s_state[SOLO_S_QUAT], uy = s_state[SOLO_S_QUAT + 1], uz = s_state[SOLO_S_QUAT + 2], uw = s_state[SOLO_S_QUAT + 3];
    m00 = T(1) - T(2) * (uy * uy + uz * uz); m01 = T(2) * (ux * uy - uw * uz); m02 = T(2) * (ux * uz + uw * uy);
    m10 = T(2) * (ux * uy + uw * uz); m11 = T(1) - T(2) * (ux * ux + uz * uz); m12 = T(2) * (uy * uz - uw * ux);
    nbr = V3<T>{T(2) * (ux * uz - uw * uy), T(2) * (uy * uz + uw * ux), T(1) - T(2) * (ux * ux + uy * uy)};
    qa1 = s_leg[leg][17]; qa2 = s_leg[leg][18];
    ck1 = s_leg[leg][19]; sk1 = s_leg[leg][20]; ck12 = s_leg[leg][21]; sk12 = s_leg[leg][22];
    p1 = V3<T>{L.hip[0], L.hip[1], L.hip[2]};
    p2 = p1 + roty(ck1, sk1, V3<T>{L.knee[0], L.knee[1], L.knee[2]});
  }
  const RowView<T> rc = tabs.row(lane);
  const int type = rc.type;
  const bool is_motor = type == ROW_MOTOR, is_contact = type >= ROW_NORMAL && type <= ROW_TAN2, is_limit = type == ROW_LIMIT;
  V3<T> cb = {rc.geo[0], rc.geo[1], rc.geo[2]};
  const T radius = rc.geo[3];
  if (rc.body == BODY_UPPER) cb = p1 + roty(ck1, sk1, cb);
  else if (rc.body == BODY_LOWER) cb = p2 + roty(ck12, sk12, cb);
  // ground under the sphere: flat plane z = 0 (plane.urdf, solo8_base_env.py:47) or the tangent
  // plane of the heightfield under the sphere centre (SoloTerrain; wave-uniform choice)
  T dist;
  V3<T> nloc = nbr;                      // ground normal in base coordinates
  V3<T> d = nbr;                         // this row's direction in base coordinates
  if (B.terrain == nullptr) {
    dist = s_state[SOLO_S_POS + 2] + dot(nbr, cb) - radius;
    if (type == ROW_TAN1) d = V3<T>{m00, m01, m02};
    if (type == ROW_TAN2) d = V3<T>{m10, m11, m12};
  } else {
    const T cwx = s_state[SOLO_S_POS] + m00 * cb.x + m01 * cb.y + m02 * cb.z;
    const T cwy = s_state[SOLO_S_POS + 1] + m10 * cb.x + m11 * cb.y + m12 * cb.z;
    const T cwz = s_state[SOLO_S_POS + 2] + nbr.x * cb.x + nbr.y * cb.y + nbr.z * cb.z;
    const T gu = (cwx - C.terr_ox) * C.terr_inv_cell, gv = (cwy - C.terr_oy) * C.terr_inv_cell;
    int gi = (int)R::floor(gu), gj = (int)R::floor(gv);
    gi = gi < 0 ? 0 : (gi > C.terr_nx - 2 ? C.terr_nx - 2 : gi);
    gj = gj < 0 ? 0 : (gj > C.terr_ny - 2 ? C.terr_ny - 2 : gj);
    const T fu = R::clamp(gu - T(gi), T(0), T(1)), fv = R::clamp(gv - T(gj), T(0), T(1));
    const T* H = B.terrain + (size_t)gj * C.terr_nx + gi;
    const T h00 = H[0], h10 = H[1], h01 = H[C.terr_nx], h11 = H[C.terr_nx + 1];
    const T hh0 = (T(1) - fu) * (T(1) - fv) * h00 + fu * (T(1) - fv) * h10 + (T(1) - fu) * fv * h01 + fu * fv * h11;
    const T hx = ((T(1) - fv) * (h10 - h00) + fv * (h11 - h01)) * C.terr_inv_cell;
    const T hy = ((T(1) - fu) * (h01 - h00) + fu * (h11 - h10)) * C.terr_inv_cell;
    const T inv = R::rsqrt(hx * hx + hy * hy + T(1));
    const V3<T> nw = {-hx * inv, -hy * inv, inv};
    // friction directions: world x projected into the tangent plane, and n x t1
    const T itn = R::rsqrt(T(1) - nw.x * nw.x);
    const V3<T> t1w = {(T(1) - nw.x * nw.x) * itn, -nw.x * nw.y * itn, -nw.x * nw.z * itn};
    const V3<T> t2w = cross(nw, t1w);
    dist = (cwz - hh0) * nw.z - radius;
    V3<T> dw = nw;
    if (type == ROW_TAN1) dw = t1w;
    if (type == ROW_TAN2) dw = t2w;
    nloc = V3<T>{m00 * nw.x + m10 * nw.y + nbr.x * nw.z, m01 * nw.x + m11 * nw.y + nbr.y * nw.z, m02 * nw.x + m12 * nw.y + nbr.z * nw.z};
    d = V3<T>{m00 * dw.x + m10 * dw.y + nbr.x * dw.z, m01 * dw.x + m11 * dw.y + nbr.y * dw.z, m02 * dw.x + m12 * dw.y + nbr.z * dw.z};
  }
  // URDF joint limit of this lane's joint ([recalled] btMultiBodyJointLimitConstraint; k = 14: HFE,
  // k = 15: KFE): distance to the NEARER limit, and the direction (+1 lower, -1 upper) that opens it
  const int ldof = k & 1;
  const T qlim = ldof == 0 ? qa1 : qa2;
  const T c_lo = qlim - L.limit[ldof][0], c_hi = L.limit[ldof][1] - qlim;
  const T lim_dir = c_lo < c_hi ? T(1) : T(-1), lim_dist = c_lo < c_hi ? c_lo : c_hi;
  const bool live = is_motor || (is_contact && dist < C.margin) || (is_limit && lim_dist < C.limit_margin);
  const V3<T> x = cb - radius * nloc;  // contact point in base coordinates
  T jb[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
  T jl1 = T(0), jl2 = T(0), bias = T(0);
  if (is_contact) {
    const V3<T> xd = cross(x, d);
    jb[0] = xd.x; jb[1] = xd.y; jb[2] = xd.z; jb[3] = d.x; jb[4] = d.y; jb[5] = d.z;
    if (rc.body != BODY_BASE) jl1 = dot(d, ycross(x - p1));
    if (rc.body == BODY_LOWER) jl2 = dot(d, ycross(x - p2));
    if (type == ROW_NORMAL) bias = (dist > T(0)) ? -dist * C.inv_dt : -C.erp_over_dt * dist;
  } else if (is_motor) {
    // POSITION_CONTROL motor row ([recalled] btMultiBodyJointMotor): target velocity
    // kp (q* - q)/dt + (1 - kd) qd, impulse clamp +-maxForce*dt
    jl1 = (k == 0) ? T(1) : T(0);
    jl2 = (k == 1) ? T(1) : T(0);
    const T qj = s_leg[leg][17 + k], uj = s_leg[leg][15 + k];
    bias = C.kp_over_dt * (my_target - qj) + C.one_minus_kd * uj;
    if constexpr (sizeof(T) == 8) target_bad = !R::finite(my_target);  // (f64: looked at where the target is used - the value does not live on to the end of the step)
  } else if (is_limit) {
    // speculative unilateral row, the same form as a contact normal row: v towards the limit
    // <= distance / dt while inside, pushed back with the erp once violated
    jl1 = (ldof == 0) ? lim_dir : T(0);
    jl2 = (ldof == 1) ? lim_dir : T(0);
    bias = (lim_dist > T(0)) ? -lim_dist * C.inv_dt : -C.erp_over_dt * lim_dist;
  }
  T gh[6], hh[2];
  {
    T g[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) g[i] = jb[i] - s_leg[leg][i] * jl1 - s_leg[leg][6 + i] * jl2;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      T s = g[i];
#pragma unroll
      for (int m = 0; m < i; ++m) s -= s_keep[i * (i - 1) / 2 + m] * gh[m];
      gh[i] = s * s_keep[15 + i];
    }
    hh[0] = jl1 * s_leg[leg][12];
    hh[1] = (jl2 - s_leg[leg][13] * hh[0]) * s_leg[leg][14];
  }
  T w = jl1 * s_leg[leg][15] + jl2 * s_leg[leg][16] - bias;
#pragma unroll
  for (int i = 0; i < 6; ++i) w += jb[i] * s_keep[21 + i];
  T diag = hh[0] * hh[0] + hh[1] * hh[1];
#pragma unroll
  for (int i = 0; i < 6; ++i) diag += gh[i] * gh[i];
  T inv_d = R::rcp(diag);
  if (!live) {
#pragma unroll
    for (int i = 0; i < 6; ++i) gh[i] = T(0);
    hh[0] = hh[1] = T(0);
    inv_d = T(0);
    w = T(0);
    diag = T(0);
  }
  // ---- warm start (SoloConfig::solver_warm_start, an OPT-IN of the residual-threshold kernels; warm_on is wave-uniform):
  //      the iteration starts from f x the impulse this row ended the robot's previous step with, clamped to this step's
  //      bounds (friction rows: mu x their contact's STARTING normal impulse).  The candidates then start at
  //      v_s = lam0_s - (w_s + (A lam0)_s) / A_ss, and A lam0 comes through the whitened vectors exactly as the post-solve
  //      phase applies impulses: Z = sum_r ghat_r lam0_r (six wave sums), Y = the same of hhat per leg, (A lam0)_s =
  //      ghat_s . Z + hhat_s . Y - no column is touched, the Gauss-Seidel loops (assembly included) only see another
  //      starting point.
  T lam0 = T(0);
  if constexpr (kResid) {
    if (warm_on) {
      if constexpr (sizeof(T) == 8) mu = s_keep[27];
      const T imp0 = C.motor_impulse;
      lam0 = live ? C.warm_factor * warm_in : T(0);
      if (is_motor) lam0 = R::clamp(lam0, -imp0, imp0);
      else if (type == ROW_NORMAL || is_limit) lam0 = R::max(lam0, T(0));
      const T wn1 = wave_lane_below<1>(lam0), wn2 = wave_lane_below<2>(lam0);  // (lane = row here: a contact's rows share a 16-lane row)
      if (type == ROW_TAN1 || type == ROW_TAN2) {
        const T lim0 = (rc.body == BODY_BASE ? s_keep[29] : mu) * (type == ROW_TAN1 ? wn1 : wn2);   // (the base link keeps its own friction: see the solver)
        lam0 = R::clamp(lam0, -lim0, lim0);
      }
      if (!live) lam0 = T(0);
      T zz[6], yy[2];
#pragma unroll
      for (int i = 0; i < 6; ++i) zz[i] = gh[i] * lam0;
      yy[0] = hh[0] * lam0;
      yy[1] = hh[1] * lam0;
      wave_reduce_rows(zz, yy);
      T alam = hh[0] * yy[0] + hh[1] * yy[1];
#pragma unroll
      for (int i = 0; i < 6; ++i) alam += gh[i] * zz[i];
      w += alam;   // (dead rows: ghat = hhat = 0, so alam = 0 and w stays 0)
    }
  }
  const unsigned long long touching = wave_ballot(live && type == ROW_NORMAL);
  const unsigned long long limited = wave_ballot(live && is_limit);
  // ---- into SOLVER space.  What a lane holds while the Gauss-Seidel iteration runs: sv_type (ROW_IDLE: nothing),
  //      its whitened row (sg, sh), the candidate at zero impulse sv_v0 = w x nid, the column scale sv_nid = -1 / A_ss,
  //      A_ss itself for the residual test; `row_at` = where THIS lane's row (the one it built above) sits in s_rowvec /
  //      s_hext, for the post-solve phase.
  //  * lane = row (f32): the lane keeps the row it built.
  //  * SLOT space (f64, ColumnBank<T>::kCompact): the live rows are permuted, in lane order, to the lanes 0 .. L-1
  //    (the dead ones, zeroed, behind them: dst is a permutation), so that the column bank needs slots for the rows
  //    that can move only (solo_wave_ops.h).  Lane order is solver order within each phase, a sphere's three rows stay
  //    neighbours (normal, tangent 1, tangent 2: live together), and the phases become lane masks of this step.  The
  //    row vectors travel through LDS, where the column build needs them anyway; the per-lane scalars through the LDS
  //    crossbar (ds_permute: no memory).
  int sv_type = live ? type : (int)ROW_IDLE;
  T sv_v0 = w * -inv_d + lam0, sv_nid = -inv_d, sv_diag = diag;  // (w = inv_d = lam0 = 0 on a dead row)
  T sv_lam0 = lam0;
  T sg[6], sh[2];
  int sv_leg = leg, n_live = 64;
  // the row's friction coefficient is its SPHERE's: the base link's spheres keep SoloConfig::base_lateral_friction - the
  // reference's changeDynamics loop sets lateralFriction for links 0 .. 11 only (solo8v2vanilla.py:157-163) -, the legs'
  // the robot's own (params[e][0]).  Only the friction rows ever read it.
  bool sv_on_base = rc.body == BODY_BASE;
  if constexpr (!kCompact) {
#pragma unroll
    for (int i = 0; i < 6; ++i) { s_rowvec[lane * kRS + i] = gh[i]; sg[i] = gh[i]; }
    s_rowvec[lane * kRS + 6] = hh[0];
    s_rowvec[lane * kRS + 7] = hh[1];
    sh[0] = hh[0]; sh[1] = hh[1];
    // the joint-space part again, in the slot of this row's leg of a [64][4 legs x 2] array whose
    // other slots stay zero: a lane reads the slot of ITS leg, so "same leg" needs no test
    s_hext[lane][2 * leg] = hh[0];
    s_hext[lane][2 * leg + 1] = hh[1];
    row_at = lane;
    wave_sync();
  } else {
    const unsigned long long live_lanes = wave_ballot(live);
    n_live = __builtin_popcountll(live_lanes);               // L (wave-uniform)
    const int below = wave_count_below(live_lanes);          // live rows on the lanes below this one
    const int dst = live ? below : n_live + (lane - below);   // a permutation of 0 .. 63
    row_at = dst;
#pragma unroll
    for (int i = 0; i < 6; ++i) s_rowvec[dst * kRS + i] = gh[i];
    // the joint-space part behind it, and the row's leg as the slot's tag (it counts only between rows of one leg)
    s_rowvec[dst * kRS + 6] = hh[0];
    s_rowvec[dst * kRS + 7] = hh[1];
    s_rowleg[dst] = (unsigned char)leg;
    const int tl = wave_push_int(sv_type | (leg << 4) | (sv_on_base ? 64 : 0), dst);
    sv_v0 = wave_push(sv_v0, dst);
    sv_nid = wave_push(sv_nid, dst);
    if constexpr (kResid) { sv_diag = wave_push(sv_diag, dst); if (warm_on) sv_lam0 = wave_push(sv_lam0, dst); }
    sv_type = tl & 15;
    sv_leg = (tl >> 4) & 3;
    sv_on_base = (tl & 64) != 0;
    wave_sync();
#pragma unroll
    for (int i = 0; i < 6; ++i) sg[i] = s_rowvec[lane * kRS + i];
    sh[0] = s_rowvec[lane * kRS + 6];
    sh[1] = s_rowvec[lane * kRS + 7];
  }

  SOLO_STAMP(B, 7);
  // ---- the scaled Delassus matrix, column by column, RESIDENT IN REGISTERS -------------------
  // Column r, as lane s sees it:  col_r[s] = -(ghat_s . ghat_r + [same leg] hhat_s . hhat_r) / A_ss
  // (0 on the row's own lane), built once per step for the rows that can move (the 8 motor rows
  // and the three rows of every touching sphere; all branches wave-uniform, all register indices
  // static, the LDS broadcasts of many columns in flight at once).  The Gauss-Seidel loop below
  // then fetches the column of the row it updates with ONE register-indexed move
  // (s_set_gpr_idx_on + v_mov) instead of an LDS round trip and a dot product on its serial chain.
  // (f64: the columns of the first ColumnBank<double>::kSlots = 32 SLOTS, 64 VGPRs; a step with more live rows
  // builds none and takes the overflow path below.)
#if SOLO_PRIO_BY_ROWS
  // Issue priority BY LIVE ROWS (round 5): the number of live rows is the best predictor of a robot-step's solver cost that
  // exists before the solve (mean sweeps 5 with one touching sphere, 20 with eight; the cost of the robot's previous steps
  // predicts next to nothing - rank correlation 0.2, profiles/round5_cost_persistence.log) - and what follows, the column
  // build (proportional to the rows) and the Gauss-Seidel iteration, is where robot-steps differ: the more rows, the higher
  // the wave's priority from here to the end of the solve (which sets the rotation / the slow robot's level again).
  // K = 20 +4 % in f64 and in f32, one launch per step and 250-step launches unchanged (profiles/round5_prio_by_rows_ab.log);
  // thresholds 8 / 14 / 20 and 14 / 20 / 26 measured the same and 3 % less.
  {
    const int rows_now = kCompact ? n_live : __builtin_popcountll(wave_ballot(live));   // (lane = row, f32: the live lanes)
    wave_set_priority_level(prio_sweeps > kPrioSweeps<T> * prio_steps ? 3 : priority_by_rows(rows_now));
  }
#endif
  ColumnBank<T> A;
  if constexpr (kCompact) A.init(sg, sh, sv_nid, lane, s_rowvec, s_rowleg, sv_leg);
  else A.init(sg, sh, sv_nid, lane, s_rowvec, &s_hext[0][2 * sv_leg]);  // (+ 8 r: row r's joint-space part if r is on this lane's leg, else 0)
  const bool overflow = kCompact && n_live > ColumnBank<T>::kSlots;  // (wave-uniform)
  if constexpr (!kCompact) {
#pragma unroll
    for (int l2 = 0; l2 < 4; ++l2) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) A.build(16 * l2 + kk);
    }
    const unsigned touching_lo = (unsigned)touching, touching_hi = (unsigned)(touching >> 32);
    if (limited != 0ull) {  // (a joint within reach of a limit is rare: one test for all eight rows)
#pragma unroll
      for (int l2 = 0; l2 < 4; ++l2) {
#pragma unroll
        for (int kk = 14; kk < 16; ++kk)
          if ((limited >> (16 * l2 + kk)) & 1ull) A.build(16 * l2 + kk);
      }
    }
#pragma unroll
    for (int l2 = 0; l2 < 4; ++l2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r0 = 16 * l2 + 2 + 3 * j;
        if (((r0 < 32 ? touching_lo : touching_hi) >> (r0 & 31)) & 1u) {  // (32-bit halves: s_bitcmp1_b32 + branch)
#pragma unroll
          for (int q = 0; q < 3; ++q) A.build(r0 + q);
        }
      }
    }
  } else if (!overflow) {
    // slots 0, 1 are the first leg's motor rows; behind them the live rows come in threes when no joint-limit row is
    // live (L = 8 + 3 x touching spheres): one test per three columns, none built in vain (a slot beyond L holds a
    // zero row: its column would be zero and is never fetched)
    // (software-pipelined by one slot: the row of slot r + 1 is fetched - eight LDS broadcasts - in front of the arithmetic
    // of slot r, across the tests too: a wave alone on its SIMD, the slow robot at the end of a launch, otherwise sits
    // out an LDS round trip at the head of every triple)
    if constexpr (SOLO_F64_WAVES >= 4 && SOLO_W4_HYBRID_BUILD && kPipelinedBuild) {
      // round 6: pipelined by ONE row while the second half of the bank is still dead.  A second row in flight (17 VGPRs) does
      // not fit next to the FULL bank (round 5: 64 spills) - but the bank fills progressively, and until slot 16 is written its
      // upper tuple (32 VGPRs) is free: the columns of slots 0 .. 16 (as far as the registers reach: three more and four values are reloaded from scratch inside the step) - all the columns of the common robot-step (8 motor rows +
      // up to three touching spheres) - are built with the next slot's row fetched behind the arithmetic of this one (a wave
      // alone on its SIMD otherwise sits out an LDS round trip per column), the slots beyond as before.
      typename ColumnBank<T>::Row cur = A.fetch(0);
      {
        typename ColumnBank<T>::Row nxt = A.fetch(1);
        A.build_from(0, cur);
        cur = A.fetch(2);
        A.build_from(1, nxt);
      }
#pragma unroll
      for (int j = 2; j < ColumnBank<T>::kSlots; j += 3) {
        if (n_live > j) {
          if (j + 2 < 17) {
            typename ColumnBank<T>::Row nxt = A.fetch(j + 1);
            A.build_from(j, cur);
            cur = A.fetch(j + 2);
            A.build_from(j + 1, nxt);
            nxt = A.fetch(j + 3);                 // (the next triple's first row; a slot beyond L holds zeros)
            A.build_from(j + 2, cur);
            cur = nxt;
          } else {
            if (j == 17) A.build_from(j, cur); else A.build(j);
            if (j + 1 < ColumnBank<T>::kSlots) A.build(j + 1);
            if (j + 2 < ColumnBank<T>::kSlots) A.build(j + 2);
          }
        }
      }
    } else if constexpr (SOLO_F64_WAVES >= 4 && !SOLO_W4_PIPELINED_BUILD) {
      // (four waves per SIMD - the A/B build: 128 VGPRs have no room for a second row in flight)
      A.build(0); A.build(1);
#pragma unroll
      for (int j = 2; j < ColumnBank<T>::kSlots; j += 3) {
        if (n_live > j) {
          A.build(j);
          if (j + 1 < ColumnBank<T>::kSlots) A.build(j + 1);
          if (j + 2 < ColumnBank<T>::kSlots) A.build(j + 2);
        }
      }
    } else {
    typename ColumnBank<T>::Row cur = A.fetch(0);
    {
      const typename ColumnBank<T>::Row n1 = A.fetch(1);
      A.build_from(0, cur);
      cur = A.fetch(2);
      A.build_from(1, n1);
    }
#pragma unroll
    for (int j = 2; j < ColumnBank<T>::kSlots; j += 3) {
      if (n_live > j) {
        const typename ColumnBank<T>::Row n1 = A.fetch(j + 1 < 64 ? j + 1 : 63);
        A.build_from(j, cur);
        if (j + 1 < ColumnBank<T>::kSlots) {
          const typename ColumnBank<T>::Row n2 = A.fetch(j + 2 < 64 ? j + 2 : 63);
          A.build_from(j + 1, n1);
          cur = A.fetch(j + 3 < 64 ? j + 3 : 63);   // (the next triple's first row; a slot beyond L holds zeros)
          if (j + 2 < ColumnBank<T>::kSlots) A.build_from(j + 2, n2);
        }
      }
    }
    }
  }
  (void)touching; (void)limited;
  SOLO_STAMP(B, 8);
  // ---- projected Gauss-Seidel, sparse in the rows that still move ---------------------------
  // Per-lane solver state: candidate v, impulse lam, bounds lo/hi (friction bounds follow their
  // contact's normal impulse), dl = clamp(v) - lam.  `pend` is the set of rows with |dl| above the
  // tolerance (solver_ulp_tolerance half-ulps of |lam|, relative) - evaluated for all 64 lanes at
  // once (v_med3, subtract, compare; the compare's lane mask IS the set).  A sweep walks the
  // pending rows in solver order (motor rows, normal rows, friction rows) with a scalar
  // find-first-set; only those rows cost anything, and after each change the set is re-evaluated,
  // so the decisions are the ones a dense sweep over every row would take.
  const bool sv_motor = sv_type == ROW_MOTOR, sv_limit = sv_type == ROW_LIMIT, sv_normal = sv_type == ROW_NORMAL;
  const bool is_tan1 = sv_type == ROW_TAN1, is_tangent = sv_type == ROW_TAN1 || sv_type == ROW_TAN2;
  const T imp = C.motor_impulse;
  if constexpr (sizeof(T) == 8) mu = s_keep[27];
  mu = sv_on_base ? s_keep[29] : mu;   // (per lane from here on: the loops take it as a vector operand; s_keep[29]: SoloConfig::base_lateral_friction, parked by the prologue)
  T lo = T(0), hi = T(0);
  if (sv_motor) { lo = -imp; hi = imp; }
  else if (sv_normal || sv_limit) hi = R::big();
  T lamv = sv_lam0;   // (0 without a warm start)
  T v = sv_v0;
  if constexpr (kResid) {
    if (warm_on) {  // the friction rows' bounds at the start: mu x the starting normal impulse of their contact
      T n1, n2;
      if constexpr (kCompact) { n1 = wave_slot_below<1>(lamv); n2 = wave_slot_below<2>(lamv); }
      else { n1 = wave_lane_below<1>(lamv); n2 = wave_lane_below<2>(lamv); }
      const T lim = mu * (is_tan1 ? n1 : n2);
      lo = is_tangent ? -lim : lo;
      hi = is_tangent ? lim : hi;
    }
  }
  const T tol_rel = T(wave_uniform(C.ulp_tol)) * R::half_ulp();
  const int iters = wave_uniform(C.iterations);  // scalar trip count
  // pybullet's solverResidualThreshold ([recalled] default 1e-7; SoloConfig::solver_residual_threshold): the iteration
  // ends after the first sweep in which max over the rows of (delta impulse x A_rr)^2 - the squared velocity-level
  // change of the row, what [recalled] btMultiBodyConstraintSolver::resolveSingleConstraintRowGeneric returns - is
  // <= the threshold.  A row is updated at most once per sweep, so its delta is lam - lam at the start of the sweep:
  // ONE test per sweep for all 64 rows.  Threshold 0 (the host default): off - the iteration runs to its fixed point.
  // It is a compile-time property of the kernel (kResid; the engine launches solo_step_kernel<T, kFull, true> when the
  // configuration asks for it): the default kernels carry none of it.
  const T resid_thr = C.resid_thr;
  // rows of one sweep, in solver order (kPhaseLanes): constants of the lane = row layout, lane masks of this step
  // in slot space
  unsigned long long ph0 = kPhaseLanes[0], ph1 = kPhaseLanes[1], ph2 = kPhaseLanes[2];
  if constexpr (kCompact) {
    ph0 = wave_ballot(sv_motor || sv_limit);
    ph1 = wave_ballot(sv_normal);
    ph2 = wave_ballot(is_tangent);
  }
  T cand = R::clamp(v, lo, hi);
  T dl = cand - lamv;
  unsigned long long pend = wave_ballot(R::abs(dl) > R::abs(lamv) * tol_rel);
  int n_changed = 0;
  int it = 0;
#ifdef SOLO_PGS_GFX950
  // on the GPU: the loop written in assembly (solo_pgs_gfx950.h, f32 and f64) - same rows, same order, same
  // arithmetic as pgs_solve_cpp, which stays the definition (the CPU emulator, and the -DSOLO_PGS_NO_ASM test
  // build the assembly is compared against bit for bit: tests/test_gpu_pgs_asm.py).
  if (!overflow) {
    const unsigned long long tan1_lanes = wave_ballot(is_tan1), tangent_lanes = wave_ballot(is_tangent);
    if constexpr (!kResid) {
      // the default configuration: ONE straight-line call runs all the sweeps
      it = pgs_solve_gfx950(A, v, lamv, cand, dl, pend, lo, hi, tol_rel, lane, mu, tan1_lanes, tangent_lanes, ph0, ph1, ph2, iters, n_changed);
    } else {
      // pybullet's residual threshold (opt-in; kernel instantiations of their own, so that the default kernels keep
      // their code): the loop is entered for ONE sweep at a time and the test - one for all 64 rows - sits between
      // the calls: ~15 instructions of entry / exit per sweep, paid only by that configuration
#pragma unroll 1
      for (;;) {
        const T lam_sweep_start = lamv;
        const int n = pgs_solve_gfx950(A, v, lamv, cand, dl, pend, lo, hi, tol_rel, lane, mu, tan1_lanes, tangent_lanes, ph0, ph1, ph2, 1, n_changed);
        it += n;
        if (n == 0 || it >= iters) break;   // (n == 0: nothing was pending at the start of the sweep)
        const T dvel = (lamv - lam_sweep_start) * sv_diag;
        if (wave_ballot(dvel * dvel > resid_thr) == 0ull) break;
      }
    }
#if defined(SOLO_PGS_HAZARD_PROBE)
    // DIAGNOSTIC builds only: is the register-index mode still ON behind the loop?  (MODE[27] = gpr_idx_en; counted in
    // slot 7 of the statistics row, and switched off so that nothing behind the loop computes with it)
    {
      const unsigned stuck = __builtin_amdgcn_s_getreg((0 << 11) | (27 << 6) | 1);
      asm volatile("s_set_gpr_idx_off\n\ts_nop 1");
      if (stuck != 0 && lane == 0) stats_add(&B.stats[7], 1.0);
    }
#endif
  } else
#else
  if (!overflow) it = pgs_solve_cpp<T, kResid, false>(A, v, lamv, cand, dl, pend, lo, hi, tol_rel, lane, mu, is_tan1, is_tangent, ph0, ph1, ph2, iters, sv_diag, resid_thr, n_changed);
  else
#endif
  {
    // the overflow path of the slot-space solver (more live rows than column slots; never taken with lane = row)
    if constexpr (kCompact) it = pgs_solve_cpp<T, kResid, true>(A, v, lamv, cand, dl, pend, lo, hi, tol_rel, lane, mu, is_tan1, is_tangent, ph0, ph1, ph2, iters, sv_diag, resid_thr, n_changed);
  }
#ifdef SOLO_STAMPS
  if (lane == 0) {
    // sweeps | touching spheres << 16 | XCC id << 24 | HW_ID[19:0] (wave, simd, pipe, cu, sh, se, tg) << 28 | row updates << 48
    const unsigned long long hw = (unsigned long long)(__builtin_amdgcn_s_getreg(63492) & 0xfffff), xcc = (unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 0xf);
    B.stamps[(size_t)B.stamp_row * 32 + 15] = (unsigned long long)(it & 0xffff) | ((unsigned long long)__builtin_popcountll(touching) << 16) | (xcc << 24) | (hw << 28) | ((unsigned long long)(n_changed & 0xffff) << 48);
#ifndef SOLO_STAMPS_LIGHT
    B.acc[15] += (unsigned long long)it;
    B.acc[0] += (unsigned long long)n_changed;
#endif
  }
#endif
  (void)n_changed;
  // back to lane = row: the impulse of the row THIS lane built (the post-solve phase sums per leg over the 16 lanes of a leg)
  if constexpr (kCompact) lamv = wave_pull(lamv, row_at);
  // Issue priority of this wave in its SIMD (s_setprio).  Two effects are countered (both measured
  // with per-wave start / end stamps, tools/gpu_tail.py):
  //  * a SIMD arbitrates its waves by priority, then AGE: at equal priority the oldest of the four
  //    resident waves issues almost unimpeded and the youngest gets the leftovers, for the whole
  //    launch (per-robot time per step 18 k ... 30 k cycles by age rank alone).  The waves therefore
  //    ROTATE through priorities 0..2 by step count and wave slot, which equalises their progress;
  //  * a launch lasts as long as its slowest robot, and slow means many Gauss-Seidel sweeps (the
  //    cost is persistent within an episode): a robot that has averaged more than kPrioSweeps (8; f64: 12)
  //    sweeps per step so far in the launch is pinned to priority 3, above the rotation.
  // +4 % on the 250-step fused rollout, +2 % at 20 steps; no effect on results.
  prio_sweeps += it;
  prio_steps += 1;
  prio_rot = prio_rot == 2 ? 0 : prio_rot + 1;  // = (prio_steps + wave slot) % 3, without the division
  wave_set_priority_level(prio_sweeps > kPrioSweeps<T> * prio_steps ? 3 : prio_rot);
  SOLO_STAMP(B, 9);
  return lamv;
#undef L
}

// post-solve half: apply the impulses (du_b = C^-T sum ghat lam ; dqd_l = Lp^-T sum hhat lam -
// K du_b), go back to world-frame velocities and integrate.  Everything is re-read from LDS.
template <typename T>
__device__ __forceinline__ void physics_finish(const StepConst<T>& C, T* s_state, T* s_scratch,
                                               const T* s_keep, const T (*s_leg)[kLegSlots], const T* s_math, T lam, int lane, int row_at) {
  const T* const s_rowvec = s_scratch;  // (the row vectors; the block they head is the scratch of the f64 reduction below)
  if constexpr (sizeof(T) == 8) lane = wave_opaque_lane(lane);  // (f64: per-lane LDS addresses are re-derived here - shared with physics_solve they lived across the whole solver, as a spill)
  using R = Real<T>;
  constexpr int kRS = ColumnBank<T>::kRowStride;
  const int leg = lane >> 4, k = lane & 15;
  const T dt = C.dt;
  // z = sum over all rows of ghat * lam (6 wave-wide sums), yl = per-leg sums of hhat * lam (2 sums over
  // the 16 lanes of a leg): eight reductions, interleaved stage by stage (wave_reduce_rows)
  T z[6], yl[2];
#pragma unroll
  for (int i = 0; i < 6; ++i) z[i] = s_rowvec[row_at * kRS + i] * lam;
  yl[0] = s_rowvec[row_at * kRS + 6] * lam;   // (the row this lane built is a row of this lane's leg)
  yl[1] = s_rowvec[row_at * kRS + 7] * lam;
  if constexpr (ColumnBank<T>::kCompact) {
    // f64: through LDS (solo_wave_ops.h: a third of the instructions of the DPP chains) - the row-vector block is the
    // scratch: every lane has just taken what it needs of it (wave_sync: the reads above come first)
    static_assert(kRowBlockReals<T> >= kReduceScratch, "the row-vector block holds the reduction's scratch");
    wave_sync();
    wave_reduce_rows_lds(z, yl, s_scratch, lane);
  } else {
    wave_reduce_rows(z, yl);
  }
  const T yl1 = yl[0], yl2 = yl[1];
  // C^T x = z (back substitution with the parked Cholesky factor)
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    T s = z[i];
#pragma unroll
    for (int m = i + 1; m < 6; ++m) s -= s_keep[m * (m - 1) / 2 + i] * z[m];
    z[i] = s * s_keep[15 + i];
  }
  const T iL11 = s_leg[leg][12], L21 = s_leg[leg][13], iL22 = s_leg[leg][14];
  const T t2 = yl2 * iL22, t1 = (yl1 - L21 * t2) * iL11;
  T kz1 = T(0), kz2 = T(0), ub[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    kz1 += s_leg[leg][i] * z[i];
    kz2 += s_leg[leg][6 + i] * z[i];
    ub[i] = s_keep[21 + i] + z[i];
  }
  const T us1 = s_leg[leg][15] + t1 - kz1;
  const T us2 = s_leg[leg][16] + t2 - kz2;
  const T q1 = s_leg[leg][17], q2 = s_leg[leg][18];
  const T qx = s_state[SOLO_S_QUAT], qy = s_state[SOLO_S_QUAT + 1], qz = s_state[SOLO_S_QUAT + 2], qw = s_state[SOLO_S_QUAT + 3];
  const T r00 = T(1) - T(2) * (qy * qy + qz * qz), r01 = T(2) * (qx * qy - qw * qz), r02 = T(2) * (qx * qz + qw * qy);
  const T r10 = T(2) * (qx * qy + qw * qz), r11 = T(1) - T(2) * (qx * qx + qz * qz), r12 = T(2) * (qy * qz - qw * qx);
  const T r20 = T(2) * (qx * qz - qw * qy), r21 = T(2) * (qy * qz + qw * qx), r22 = T(1) - T(2) * (qx * qx + qy * qy);

  // ---- back to world-frame velocities, integrate positions ----------------------------------
  const V3<T> wn = {r00 * ub[0] + r01 * ub[1] + r02 * ub[2], r10 * ub[0] + r11 * ub[1] + r12 * ub[2], r20 * ub[0] + r21 * ub[1] + r22 * ub[2]};
  const V3<T> vn = {r00 * ub[3] + r01 * ub[4] + r02 * ub[5], r10 * ub[3] + r11 * ub[4] + r12 * ub[5], r20 * ub[3] + r21 * ub[4] + r22 * ub[5]};
  // q+ = exp(dt w / 2) (x) q, renormalised
  // exp(dt w / 2) = (w sin(x) / |w|, cos(x)) with x = |w| dt / 2: sin(x) / |w| = (dt / 2) sinc(x)
  T sinc_x, ch;
  R::sinc_cos(T(0.25) * dt * dt * dot(wn, wn), &sinc_x, &ch, s_math);  // argument: x^2
  const T sc = T(0.5) * dt * sinc_x;
  const T dx = wn.x * sc, dy = wn.y * sc, dz = wn.z * sc, dw = ch;
  T nx = dw * qx + dx * qw + dy * qz - dz * qy;
  T ny = dw * qy - dx * qz + dy * qw + dz * qx;
  T nz = dw * qz + dx * qy - dy * qx + dz * qw;
  T nw = dw * qw - dx * qx - dy * qy - dz * qz;
  const T nn = R::rsqrt(nx * nx + ny * ny + nz * nz + nw * nw);
  wave_sync();  // every lane has finished reading the old state
  if (lane == 0) {
    s_state[SOLO_S_POS] += dt * vn.x; s_state[SOLO_S_POS + 1] += dt * vn.y; s_state[SOLO_S_POS + 2] += dt * vn.z;
    s_state[SOLO_S_QUAT] = nx * nn; s_state[SOLO_S_QUAT + 1] = ny * nn; s_state[SOLO_S_QUAT + 2] = nz * nn; s_state[SOLO_S_QUAT + 3] = nw * nn;
    s_state[SOLO_S_ANGVEL] = wn.x; s_state[SOLO_S_ANGVEL + 1] = wn.y; s_state[SOLO_S_ANGVEL + 2] = wn.z;
    s_state[SOLO_S_LINVEL] = vn.x; s_state[SOLO_S_LINVEL + 1] = vn.y; s_state[SOLO_S_LINVEL + 2] = vn.z;
  }
  if (k == 0) {
    s_state[SOLO_S_Q + 2 * leg] = q1 + dt * us1;
    s_state[SOLO_S_Q + 2 * leg + 1] = q2 + dt * us2;
    s_state[SOLO_S_QD + 2 * leg] = us1;
    s_state[SOLO_S_QD + 2 * leg + 1] = us2;
  }
  wave_sync();
}

// ------------------------------------------------------------------------------------------
// the fused kernel
// ------------------------------------------------------------------------------------------
// __launch_bounds__(64, W): W waves per SIMD -> 512/W VGPRs.  f32: 4 (128 VGPRs, a whole 4096-robot
// batch resident on the 1024 SIMDs); f64: 3 (168 VGPRs - the resident columns of the slot-space solver are 64 - and
// 13.0 KB of LDS: twelve workgroups per CU; round 3: 2, with 128 VGPRs of columns).
// kFull = false: physics-only instantiation (flags are treated as SOLO_STEP_PHYSICS).
// single-step launches evaluate their outputs in the step kernel itself (see the step loop): f32 only
template <typename T, bool kFull> constexpr bool kInlineOutputs = kFull;
template <typename T> constexpr int kWavesPerSimd = sizeof(T) == 4 ? 4 : SOLO_F64_WAVES;

template <typename T, bool kFull, bool kResid = false, bool kMigrate = false>
__global__ __launch_bounds__(64, kWavesPerSimd<T>) void solo_step_kernel(const KParams<T>* __restrict__ Pin, KBuffers<T> Bin) {
  KBuffers<T> B = Bin;
  if (!kFull) B.flags = SOLO_STEP_PHYSICS;
  using R = Real<T>;
  __shared__ T s_state[SOLO_STATE_STRIDE];
  // ONE block: the whitened row vectors [64][kRS], their joint-space parts by leg slot [64][4 legs x 2] (see
  // physics_solve; zero except the row's own leg) and the per-row geometry [64][centre 3, radius] - the output epilogue
  // (where all three are dead) uses it as its 1024-value scratch
  constexpr int kRS = ColumnBank<T>::kRowStride;
  constexpr int kRowsReals = kRowBlockReals<T>;
  // (the geometry table has one entry per SPHERE - a contact's three rows share it - and an all-zero entry for the rows
  // that have none: as [64 rows][4] it was 2 KB of the f64 kernel's 13.2 KB, and 13.2 KB round up to eleven LDS
  // allocation granules of 1280 B: ELEVEN workgroups per CU where the registers allow twelve - round 5)
  constexpr int kGeoRows = SOLO_MAX_SPHERES + 1;
  __shared__ T s_blk[kRowsReals + kGeoRows * 4];
  static_assert(kRowsReals >= SOLO_MAX_REWARD_OPS * (kRowsReals / SOLO_MAX_REWARD_OPS < 32 ? kRowsReals / SOLO_MAX_REWARD_OPS : 32), "the output epilogue's scratch");
  T* const s_rowvec = s_blk;
  T (*const s_hext)[8] = ColumnBank<T>::kCompact ? nullptr : reinterpret_cast<T (*)[8]>(s_blk + 64 * kRS);
  __shared__ unsigned char s_rowleg[64];  // (slot space: the leg of every slot's row)
  T (*const s_rowgeo)[4] = reinterpret_cast<T (*)[4]>(s_blk + kRowsReals);
  __shared__ int32_t s_rowtype[64];  // (StepTables::rowtype)
  __shared__ T s_keep[32];
  __shared__ T s_leg[4][kLegSlots];
  // termination (termination.py:38-83), one lane per termination (lanes >= SOLO_MAX_TERMS: never fire):
  // s_cnt = TimeBased step counters, s_termlim = the count above which lane t fires (-1: always - a
  // Constant(True) -, INT_MAX: never), s_termtick = 1 for the lanes whose counter ticks (TimeBased)
  // (SOLO_MAX_TERMS entries each: as [64] they were 768 B for four live entries)
  __shared__ int s_cnt[SOLO_MAX_TERMS];
  __shared__ int s_termlim[SOLO_MAX_TERMS];
  __shared__ int s_termtick[SOLO_MAX_TERMS];
  // per-lane constant tables, staged ONCE per launch (a launch fuses many steps): the steps then
  // read them from LDS instead of paying a global-load latency each
  __shared__ LegConst<T> s_legc[4];
  __shared__ StepConst<T> s_const;          // the scalars a step reads (see solo_kernel_params.h)
  // coefficient table of Real<T>'s polynomials (f64 only: see Real<double>::sincos; f32 uses instruction literals)
  __shared__ T s_math[Real<T>::kTabSize > 0 ? Real<T>::kTabSize : 1];

  const int lane0 = lane_id();
  const int slot = block_id() + B.env_base;
  if (slot >= B.num_envs) return;
  // (wave_cold_args assumes the kernel's parameter layout - one pointer, then this block: checked on
  // two fields, so that a changed signature traps instead of reading garbage)
  if (wave_cold_args(Bin)->num_envs != B.num_envs || wave_cold_args(Bin)->steps != B.steps) __builtin_trap();
  // workgroup -> robot: the cost-balanced launch order if one is set (solo_engine_set_order: dispatch position ->
  // robot), else the XCD-contiguous map (xcd_contiguous, solo_kernel_params.h: the robots whose waves share an L2 are
  // neighbours in the batch, so their rows of the [step][robot][.] arrays complete each other's cache lines there)
  const int32_t* order = wave_cold_args(Bin)->order;
  // A TASK = one robot and a range of the launch's steps.  kMigrate = false: this workgroup's robot, all steps.
  // kMigrate (SoloConfig::migrate_steps; the queue: solo_kernel_params.h): chunks of q_chunk steps of whichever robot
  // is ready next, taken from the ring of this wave's XCD first - the wave loops over tasks until the rings hold no
  // ticket, and a robot moves from wave to wave as its record in device memory.  A kernel instantiation of its own:
  // the one-robot-per-wave kernels keep their straight-line code.
  int32_t* const queue = kMigrate ? wave_cold_args(Bin)->queue : nullptr;
  int env = 0, step_begin = 0, step_end = B.steps;
  bool last_chunk = true;   // this task ends the robot's launch: output epilogue, final bookkeeping
  int q_ring = 0, q_rings_left = 0, q_sweeps = 0, q_chunk_at = 0;
  int next_ticket = 0;      // the next task's ticket, taken with the publication of the previous one (see the end of the task loop)
  bool have_next = false;
  if constexpr (!kMigrate) env = order != nullptr ? wave_uniform(order[slot]) : B.env_base + xcd_contiguous(block_id(), B.count);
  else {
    // home ring: one of the rings of this wave's XCD (q_rings = 8 x rings per XCD, or 1)
    const int per_xcd = B.q_rings >= 8 ? B.q_rings >> 3 : 1;
    q_ring = B.q_rings >= 8 ? (wave_xcc_id() & 7) * per_xcd + (block_id() >> 3) % per_xcd : 0;
    q_rings_left = B.q_rings;
  }
#ifdef SOLO_STAMPS
  B.stamp_row = env;
#ifndef SOLO_STAMPS_LIGHT   // (the light build keeps the product's LDS footprint - 10240 B in f64: sixteen workgroups per CU - and its residency)
  __shared__ unsigned long long s_acc[17];
  if (lane0 < 17) s_acc[lane0] = lane0 == 16 ? __builtin_amdgcn_s_memtime() : 0ull;
  B.acc = s_acc;
  wave_sync();
#endif
#endif
  SOLO_STAMP(B, 0);
  // episodic statistics are sharded over SOLO_STATS_SHARDS rows: all robots of a batch finish
  // their episodes in the same step, and same-address atomics serialise at ~12 ns each
  // (the rarely touched buffers are re-read from the kernarg segment where they are used: wave_cold_args)
#define SOLO_STATS_ROW (wave_cold_args(Bin)->stats + (size_t)(env % SOLO_STATS_SHARDS) * SOLO_STATS_WIDTH)

  const KParams<T>* __restrict__ const P0 = Pin;
  // ---- the per-launch tables (per-leg / per-row / per-step constants, the polynomial coefficients): loaded ...
  constexpr int kLegWords = (int)(sizeof(LegConst<T>) * 4 / sizeof(T)), kLegLoads = (kLegWords + 63) / 64;
  constexpr int kConstWords = (int)(sizeof(StepConst<T>) / sizeof(int32_t)), kConstLoads = (kConstWords + 63) / 64;
  T leg_w[kLegLoads];
  int32_t const_w[kConstLoads];
  RowConst<T> row_w;
  T math_w = T(0);
  auto load_tables = [&]() {
    const T* leg_src = reinterpret_cast<const T*>(P0->leg);
    const int32_t* const_src = reinterpret_cast<const int32_t*>(&P0->c);
#pragma unroll
    for (int j = 0; j < kLegLoads; ++j) leg_w[j] = (lane0 + 64 * j < kLegWords) ? leg_src[lane0 + 64 * j] : T(0);
    row_w = P0->row[lane0];
#pragma unroll
    for (int j = 0; j < kConstLoads; ++j) const_w[j] = (lane0 + 64 * j < kConstWords) ? const_src[lane0 + 64 * j] : 0;
    if constexpr (Real<T>::kTabSize > 0) math_w = wave_math_table<T>(lane0 < Real<T>::kTabSize ? lane0 : 0);
  };
  // ... and staged into LDS
  auto store_tables = [&]() {
    T* leg_dst = reinterpret_cast<T*>(s_legc);
    int32_t* const_dst = reinterpret_cast<int32_t*>(&s_const);
#pragma unroll
    for (int j = 0; j < kLegLoads; ++j) if (lane0 + 64 * j < kLegWords) leg_dst[lane0 + 64 * j] = leg_w[j];
    const bool has_geo = row_w.type >= ROW_NORMAL && row_w.type <= ROW_TAN2;   // (row_w.dof: the row's model sphere)
    s_rowtype[lane0] = row_w.type | (row_w.body << 4) | ((has_geo ? row_w.dof : SOLO_MAX_SPHERES) << 8) | (leg_sum_entry(lane0 < 27 ? lane0 : 0) << 16);
    if (row_w.type == ROW_NORMAL) {
#pragma unroll
      for (int i = 0; i < 3; ++i) s_rowgeo[row_w.dof][i] = row_w.center[i];
      s_rowgeo[row_w.dof][3] = row_w.radius;
    }
    if (lane0 < 4) s_rowgeo[SOLO_MAX_SPHERES][lane0] = T(0);
#pragma unroll
    for (int j = 0; j < kConstLoads; ++j) if (lane0 + 64 * j < kConstWords) const_dst[lane0 + 64 * j] = const_w[j];
    if constexpr (Real<T>::kTabSize > 0) { if (lane0 < Real<T>::kTabSize) s_math[lane0] = math_w; }
  };
  // the termination tables, from the staged constants
  auto make_term_tables = [&]() {
    const int tl = lane0 & (SOLO_MAX_TERMS - 1);
    const int kind = s_const.term_kind[tl], param = s_const.term_param[tl];
    const bool mine = lane0 < s_const.num_terms;  // (num_terms <= SOLO_MAX_TERMS)
    if (lane0 < SOLO_MAX_TERMS) {
      s_termlim[lane0] = (mine && kind == SOLO_T_TIME) ? param : ((mine && kind == SOLO_T_CONST && param != 0) ? -1 : 0x7fffffff);
      s_termtick[lane0] = (mine && kind == SOLO_T_TIME) ? 1 : 0;
    }
  };
  if constexpr (kMigrate) {  // once per wave, in front of the task loop
    load_tables();
    store_tables();
    wave_sync();
    make_term_tables();
  }
  do {  // ---- the task loop (kMigrate; else one pass)
  if constexpr (kMigrate) {
    // a ticket of the current ring; a ring without tickets sends the wave on to the next one, and a whole round of
    // empty rings ends it
    const int chunks = migration_chunks(B.steps, B.q_chunk), per_ring = B.count / B.q_rings, ring_len = per_ring * chunks;
    int32_t* const ring_slots = queue + kQueueHeader + (size_t)B.count;
    // (a ticket is taken when the wave is FREE - at the earliest together with the publication of its previous task,
    // below -, never while it still works: a ticket reserved during the last step of a task is matched with a robot in
    // reservation order, not in the order waves become free, and waves then wait for "their" robot while others are
    // ready - measured: slower at every chunk length)
    int ticket = ring_len;
    bool tried = false;
    if (have_next) { ticket = next_ticket; tried = true; have_next = false; }
    for (;;) {
      if (ticket < ring_len) break;
      if (tried) { if (--q_rings_left <= 0) break; q_ring = q_ring + 1 == B.q_rings ? 0 : q_ring + 1; }
      // (the ring the wave is on: one read-modify-write - one device-scope round trip; a ring it walks on to at the
      // end of a launch is first looked at with a load: read-modify-writes of one address serialise at ~12 ns each,
      // and every wave ends by walking over every ring)
      if (lane0 == 0) {
        ticket = tried ? wave_atomic_load(queue + q_ring * 32) : 0;
        if (ticket < ring_len) ticket = wave_atomic_add(queue + q_ring * 32, 1);
      }
      ticket = wave_readlane_int(ticket, 0);
      tried = true;
    }
    if (ticket >= ring_len) break;
    // the slot of that ticket: published already unless more waves ask than robots are ready (the end of a launch).
    // BOUNDED wait: a wave that gives up counts itself in slot 6 of the statistics and leaves (never observed; a
    // launch must not hang on a bug)
    int ready = -1;
    for (int spin = 0; spin < SOLO_QUEUE_SPINS; ++spin) {
      if (lane0 == 0) ready = wave_atomic_load(ring_slots + (size_t)q_ring * ring_len + ticket);
      ready = wave_readlane_int(ready, 0);
      if (ready >= 0) break;
      wave_backoff();
    }
    if (ready < 0) {   // (the host finds the word set at its next call: SOLO_ERR_INCOMPLETE)
      if (lane0 == 0) { stats_add(&wave_cold_args(Bin)->stats[6], 1.0); if (wave_cold_args(Bin)->fault != nullptr) wave_fault_set(wave_cold_args(Bin)->fault); }
      break;
    }
    wave_acquire_device();  // (orders the loads of the robot's record and counters behind the poll)
    // the slot says which robot and which of its chunks: everything else is loaded in ONE round trip below
    env = B.env_base + (ready & 0xffffff);
    q_chunk_at = ready >> 24;
    step_begin = q_chunk_at * B.q_chunk;
    step_end = step_begin + B.q_chunk < B.steps ? step_begin + B.q_chunk : B.steps;
    last_chunk = step_end == B.steps;
  }
  const size_t rec = (size_t)env * SOLO_STATE_STRIDE;
  // (lane = row keeps the joint-space parts of dead legs' slots at zero - they are written once: here, and again after
  // an output epilogue has used the block as scratch; slot space rewrites all eight every step)
  if constexpr (!ColumnBank<T>::kCompact) {
#pragma unroll
    for (int i = 0; i < 8; ++i) s_hext[lane0][i] = T(0);
  }
  // ---- the prologue's global loads, ALL ISSUED BEFORE THE FIRST ONE IS WAITED FOR (written as copy loops
  //      and load-then-store pairs they were eight exposed round trips to memory, one after the other:
  //      nothing in a fused launch, 17 % of a closed-loop step, which is a launch of its own)
  if constexpr (!kMigrate) load_tables();
  T state_w = T(0);
  int count_w = 0;
  if constexpr (kMigrate) {  // (what a robot travels as is read and written with device-coherent accesses: solo_wave_ops.h)
    if (lane0 < SOLO_STATE_STRIDE) state_w = wave_load_shared(wave_cold_args(Bin)->state + rec + lane0);
    if (lane0 < SOLO_MAX_TERMS) count_w = wave_atomic_load(wave_cold_args(Bin)->term_count + (size_t)env * SOLO_MAX_TERMS + lane0);
  } else {
    if (lane0 < SOLO_STATE_STRIDE) state_w = wave_cold_args(Bin)->state[rec + lane0];
    if (lane0 < SOLO_MAX_TERMS) count_w = wave_cold_args(Bin)->term_count[(size_t)env * SOLO_MAX_TERMS + lane0];
  }
  const T mu = wave_cold_args(Bin)->params[(size_t)env * 4 + 0];
  const T mass_scale = wave_cold_args(Bin)->params[(size_t)env * 4 + 1];
  const T mu_base = P0->mu_base;
  // issue priority (see physics_solve): a closed-loop step() is a launch of ONE step - it has no history
  // of its own, and its slowest robot, one that runs all the sweeps, decides how long the step takes.  A
  // robot's Gauss-Seidel cost is persistent, so such a launch starts from the sweep count of the robot's
  // previous step (fused launches build their own history: seeded the same way they were 4 % slower)
  int hist_w = 0, prio_steps = 0;
  if (B.steps == 1) {
    const int32_t* cost = wave_cold_args(Bin)->cost;
    if ((B.flags & SOLO_STEP_PHYSICS) && cost != nullptr) { hist_w = cost[env]; prio_steps = 1; }
  }
  if constexpr (kMigrate) {  // (a migrating robot brings its history along: its sweeps so far in this launch)
    if (lane0 == 0) hist_w = wave_atomic_load(queue + kQueueHeader + (env - B.env_base));
    hist_w = wave_readlane_int(hist_w, 0);
    prio_steps = step_begin;
  }
  // ---- ... and into LDS: the per-leg / per-row / per-step tables, the state record, the TimeBased counters
  //      (kept in scalar registers next to the termination program they cost 25 SGPR spills in the step loop)
  {
    if constexpr (!kMigrate) store_tables();
    if (lane0 < SOLO_STATE_STRIDE) s_state[lane0] = state_w;
    if (lane0 < SOLO_MAX_TERMS) s_cnt[lane0] = count_w;
    // f64: the robot's friction coefficient and base-mass scale wait in LDS, not in two register pairs held across the
    // whole step loop (the f64 kernel lives on 168 VGPRs: see physics_solve, "PARK EARLY")
    if constexpr (sizeof(T) == 8) { if (lane0 == 0) { s_keep[27] = mu; s_keep[28] = mass_scale; } }
    if (lane0 == 0) s_keep[29] = mu_base;   // (the base link's own friction coefficient: physics_solve)
  }
  int prio_sweeps = wave_uniform(hist_w);
  const int hist_sweeps = kMigrate ? 0 : prio_sweeps;
  int prio_rot = (prio_steps + wave_slot_id()) % 3;  // the rotation's phase (advanced once per step)
  // (a migrating robot's history is its sweeps over the prio_steps steps it has behind it in this launch; a single-step
  // launch's the sweeps of the robot's previous step)
  if (prio_steps > 0) wave_set_priority_level(prio_sweeps > kPrioSweeps<T> * (kMigrate ? prio_steps : 1) ? 3 : wave_slot_id() % 3);
  wave_sync();
  if constexpr (!kMigrate) make_term_tables();
  // The auto-reset belongs to a step that advanced the simulation (or asks for it explicitly): a
  // query-only launch - TerminationFactory.is_terminated() outside step(), termination.py:38-50 - never
  // mutates the physics state.
  const bool may_restart = (B.flags & SOLO_STEP_DONE) && (B.flags & (SOLO_STEP_PHYSICS | SOLO_STEP_AUTO_RESET)) &&
                           wave_uniform(s_const.auto_reset) != 0;

  // B.steps consecutive env steps of THIS robot in one launch: the state record stays in LDS,
  // only actions come in and the step records / done flags go out per step.  Robots are independent, so
  // no wave ever waits for another one; a launch lasts as long as its slowest robot's SUM over
  // the steps, which averages out the contact-count imbalance between robots.
  wave_sync();  // the staged tables, the state record and the counters are in LDS
#pragma unroll 1
  for (int step = step_begin; step < step_end; ++step) {
    const StepConst<T>& C = s_const;  // (LDS: re-read every step, nothing carried across the step loop in registers)
    // per-lane address arithmetic stays in the step instead of being hoisted out of the fused step loop and kept live
    // across it (spills).  kLean (f64: the kernel lives on exactly 168 VGPRs, and what it spilled was reloaded from
    // scratch INSIDE the step - behind an s_waitcnt vmcnt(0) that also waited for the step's freshly issued action
    // load): the lane number itself is computed here (solo_wave_ops.h: wave_fresh_lane), "are there actions?" is a compare
    // on two scalar registers here instead of a flag parked in a vector register across the loop (only the test is
    // opaque: through an opaque pointer the loads became flat loads), the target's finiteness is looked at where the
    // target is used (physics_solve) instead of keeping it to the end of the step, and physics_finish re-derives its
    // LDS addresses.  f32 (0 spills without any of it, 1 % slower with it) keeps its code.
    constexpr bool kLean = sizeof(T) == 8;
    int lane = kLean ? wave_fresh_lane() : wave_opaque_lane(lane0);
    const bool have_actions = kLean ? wave_opaque_bits((unsigned long long)B.actions) != 0ull : B.actions != nullptr;
    const T* const actions = have_actions ? B.actions : nullptr;
    const StepTables<T> tabs = {s_legc, s_rowtype, s_rowgeo};
    // setJointMotorControlArray (solo8v2vanilla.py:87-90): every motor lane fetches the target of ITS
    // joint straight from global memory.  The value is consumed when the motor rows are built,
    // thousands of cycles into the step, so the load's latency is never waited for (funnelled
    // through LDS at the top of the step - or prefetched a step ahead into a register the compiler
    // then copies at once - it cost an exposed global-memory round trip per step).
    // (f64, round 5: the load is issued INSIDE physics_solve, behind the leg phase - the step's register peak - and still
    // ~3000 cycles in front of its use; at the top of the step its register pair was the first thing the 128-VGPR kernel
    // spilled)
    bool motor_lane = (s_rowtype[lane] & 15) == ROW_MOTOR;
    // (the robot's motor targets are written from several chunks - the last step's action, an auto-reset's settle pose -
    // and read back when a launch brings no actions: device-coherent accesses in a migrating launch, like its record)
    auto fetch_target = [&](int ln) -> T {   // action de-normalisation (solo8v2vanilla.py:84-85) included
      const size_t tgt_at = (size_t)env * SOLO_NUM_JOINTS + (size_t)(3 * (ln >> 4) + (ln & 15));  // pybullet joint index
      T raw_target = T(0);
      if ((ln & 15) < 2) {   // (the motor rows: k = 0, 1 of every leg - solo_kernel_params.h)
        if (actions != nullptr) raw_target = actions[(size_t)step * B.action_stride + tgt_at];
        else if constexpr (kMigrate) raw_target = wave_load_shared(wave_cold_args(Bin)->targets + tgt_at);
        else raw_target = wave_cold_args(Bin)->targets[tgt_at];
      }
      return raw_target * (actions != nullptr ? s_const.action_scale : T(1));
    };
    if (actions != nullptr && step == B.steps - 1 && lane < SOLO_NUM_JOINTS) {  // the view's targets: all 12 entries
      const T tv = actions[(size_t)step * B.action_stride + (size_t)env * SOLO_NUM_JOINTS + lane] * C.action_scale;
      if constexpr (kMigrate) wave_store_shared(wave_cold_args(Bin)->targets + (size_t)env * SOLO_NUM_JOINTS + lane, tv);
      else wave_cold_args(Bin)->targets[(size_t)env * SOLO_NUM_JOINTS + lane] = tv;
    }

    // the warm-start cache (SoloConfig::solver_warm_start; residual-threshold kernels only): this lane's row's impulse at
    // the end of the robot's previous step, fetched now and used when the rows are built
    T* const warm_row = kResid ? wave_cold_args(Bin)->warm : nullptr;   // (wave-uniform; null = off)
    T warm_in = T(0);
    if constexpr (kResid) if (warm_row != nullptr && (B.flags & SOLO_STEP_PHYSICS)) {
      if constexpr (kMigrate) warm_in = wave_load_shared(warm_row + (size_t)env * 64 + lane);
      else warm_in = warm_row[(size_t)env * 64 + lane];
    }
    SOLO_STAMP(B, 1);
    bool diverged = false;
    if (B.flags & SOLO_STEP_PHYSICS) {
      const T my_target = kLean ? T(0) : fetch_target(lane);
      bool target_bad = false;  // (set on a motor lane whose target is not finite)
      int row_at;  // where this lane's constraint row sits in s_rowvec / s_hext (its lane, or its slot: see physics_solve)
      // (the pipelined column build: the default-solver kernels whose robots do not migrate - the others, with a value or two more
      // live across the step, would reload them from scratch inside it)
      const T lam = physics_solve<T, kResid, !kResid && !kMigrate>(C, B, tabs, s_state, my_target, fetch_target, s_rowvec, s_hext, s_rowleg, s_keep, s_leg, s_math, mu, mass_scale, lane, row_at, target_bad, prio_sweeps, prio_steps, prio_rot,
                                             warm_in, kResid && warm_row != nullptr);
      if constexpr (kResid) if (warm_row != nullptr) {
        if constexpr (kMigrate) wave_store_shared(warm_row + (size_t)env * 64 + lane, lam);
        else warm_row[(size_t)env * 64 + lane] = lam;
      }
      if constexpr (kLean) lane = wave_fresh_lane();   // (nothing lane-derived lives across physics_solve)
      physics_finish<T>(C, s_state, s_rowvec, s_keep, s_leg, s_math, lam, lane, row_at);
      // a robot whose state went non-finite - or that was handed a non-finite target, which the
      // solver's clamps would otherwise swallow silently - is restored from its snapshot and counted
      const bool bad = (lane < SOLO_S_RETURN && !R::finite(s_state[lane & 31])) || (kLean ? target_bad : (motor_lane && !R::finite(my_target)));
      diverged = wave_ballot(bad) != 0ull;
      if (diverged) {
        if constexpr (kResid) if (warm_row != nullptr) {  // (a restored robot starts from zero impulses)
          if constexpr (kMigrate) wave_store_shared(warm_row + (size_t)env * 64 + lane, T(0));
          else warm_row[(size_t)env * 64 + lane] = T(0);
        }
        if (lane < SOLO_S_RETURN) s_state[lane] = wave_cold_args(Bin)->snapshot[rec + lane];
        if (lane == 0) stats_add(&SOLO_STATS_ROW[5], 1.0);
        wave_sync();
      }
    }

    SOLO_STAMP(B, 10);
    // ---- termination: OR with short-circuit, per-env TimeBased counters (termination.py:38-83).
    //      Lane t evaluates termination t on its own counter; once an earlier termination fires, the
    //      later ones are not ticked (termination.py:46-48).  Branch-free: ~14 instructions.
    bool done = false;
    if (B.flags & SOLO_STEP_DONE) {
      const int tl = lane & (SOLO_MAX_TERMS - 1);
      const bool term_lane = lane < SOLO_MAX_TERMS;
      const int old = s_cnt[tl];
      const unsigned long long fired = wave_ballot(term_lane && old + 1 > s_termlim[tl]);
      done = fired != 0ull;
      const int first = done ? __builtin_ctzll(fired) : 63;  // wave-uniform
      if (term_lane) s_cnt[tl] = old + ((s_termtick[tl] != 0 && lane <= first) ? 1 : 0);
    }
    const bool restart = may_restart && (done || diverged);
    // ---- the step's record for the output epilogue (end of this kernel): the state after the step, before
    //      an auto-reset, as ONE coalesced 32-real store; slot 31 carries the step's event bits (the
    //      epilogue turns them into the done flags and the episodic bookkeeping: no byte stores here)
    if (B.traj != nullptr) {
      const T ev = T((done ? kEventDone : 0) | (restart ? kEventRestart : 0));
      const T word = s_state[lane & (SOLO_STATE_STRIDE - 1)];
      if (lane < SOLO_STATE_STRIDE)
        B.traj[(unsigned)((env - B.env_base) * B.steps + step) * (unsigned)SOLO_STATE_STRIDE + (unsigned)lane] = lane == SOLO_S_SPARE ? ev : word;
    }
    // closed-loop step() = a single-step launch: its outputs are evaluated right here with the
    // same per-item functions the output epilogue uses (no second launch on the critical path of a
    // policy loop) - lane i takes observation element i / reward leaf i, lane 0 folds the reward
    // (f32 only: in f64 - the parity path - every launch leaves records for the output epilogue.  The library
    // atan2 / asin / exp of the f64 outputs need ~40 f64 constants, which the compiler kept live across the whole
    // step loop - and spilled: 36 scratch stores per lane at the top of every launch, 900 B of HBM writes per
    // env-step of a 20-step launch - for a code path fused launches never take.)
    if constexpr (kInlineOutputs<T, kFull>) if (B.obs_inline != nullptr || B.reward_inline != nullptr) {
      // lane i's observation element / reward instruction come from the parameter block in global memory:
      // loaded HERE so that the loads fly while the Euler angles are computed
      // (loaded where they are used they were three exposed round trips at the end of every closed-loop step)
      const int n_obs = wave_uniform(C.num_obs), n_rops = wave_uniform(C.num_reward_ops);
      // (lanes beyond a program load its entry 0 - one more address in an already issued load - and never use it)
      const ObsElemK<T> prog_obs = P0->obs[lane < n_obs ? lane : 0];
      const RewardInstrK<T> prog_reward = P0->reward[lane < n_rops ? lane : 0];
      // the three Euler angles on three LANES, one atan2 for all of them (solo_outputs.h: euler_component - the function the
      // output epilogue calls per angle, so the two paths agree bit for bit), broadcast to the wave
      const T angle = euler_component<T>(lane < 3 ? lane : 0, s_state[SOLO_S_QUAT], s_state[SOLO_S_QUAT + 1], s_state[SOLO_S_QUAT + 2], s_state[SOLO_S_QUAT + 3]);
      const T roll = wave_readlane(angle, 0), pitch = wave_readlane(angle, 1), yaw = wave_readlane(angle, 2);
      if (B.obs_inline != nullptr && lane < n_obs)
        B.obs_inline[(size_t)env * n_obs + lane] = observation_value<T>(prog_obs, s_state, roll, pitch, yaw);
      if (B.reward_inline != nullptr) {
        // lane i holds instruction i and its value: the leaves are evaluated lane-parallel, the
        // combining instructions (SCALE / ADD / MUL over earlier values, three-address form) in
        // program order with wave-uniform v_readlane broadcasts - no LDS, and no chain of dependent
        // scalar loads of the program on lane 0 (~14 x 250 cycles at the end of every closed-loop step)
        const RewardInstrK<T>& ri = prog_reward;  // (lanes >= n_rops hold a copy of instruction 0: evaluated, never read)
        T myval = reward_is_leaf(ri.op) ? reward_leaf<T>(ri, s_state, roll, pitch) : T(0);
        for (int i = 0; i < n_rops; ++i) {
          const int op = wave_readlane_int(ri.op, i);
          if (reward_is_leaf(op)) continue;                      // (wave-uniform)
          const int src = wave_readlane_int(ri.src, i);
          const T x0 = wave_readlane(myval, src & 255), x1 = wave_readlane(myval, (src >> 8) & 255);
          const T res = op == SOLO_R_SCALE ? wave_readlane(ri.a, i) * x0 : (op == SOLO_R_ADD ? x0 + x1 : x0 * x1);
          myval = (lane == i) ? res : myval;
        }
        const T reward_value = wave_readlane(myval, n_rops - 1);
        if (lane == 0) {
          const T r = reward_value;
          B.reward_inline[env] = r;
          if (B.flags & SOLO_STEP_DONE) {
            // episodic return / length live in the record's slots 29, 30: loaded with the state in the
            // prologue, updated here in LDS, stored with the state at the end of the launch
            const uint8_t ev = (uint8_t)((done ? kEventDone : 0) | (restart ? kEventRestart : 0));
            accumulate_returns<T>(s_state, &ev, 0, &r, 0, 1, SOLO_STATS_ROW, [](double* p, double x) { stats_add(p, x); });
          }
        }
      }
    }
    SOLO_STAMP(B, 11);
    if (B.flags & SOLO_STEP_DONE) {
      if (restart) {
        wave_sync();  // the record above is read from the old state first
        if (lane < SOLO_S_RETURN) s_state[lane] = wave_cold_args(Bin)->snapshot[rec + lane];
        if (lane < SOLO_MAX_TERMS) s_cnt[lane] = 0;
        if constexpr (kResid) if (warm_row != nullptr) {  // (... and so does a robot that starts a new episode)
          if constexpr (kMigrate) wave_store_shared(warm_row + (size_t)env * 64 + lane, T(0));
          else warm_row[(size_t)env * 64 + lane] = T(0);
        }
        // reset() leaves the motors commanded to the settle pose (solo8v2vanilla.py:127-136)
        if (lane < SOLO_NUM_JOINTS) {
          if constexpr (kMigrate) wave_store_shared(wave_cold_args(Bin)->targets + (size_t)env * SOLO_NUM_JOINTS + lane, C.settle_tgt[lane]);
          else wave_cold_args(Bin)->targets[(size_t)env * SOLO_NUM_JOINTS + lane] = C.settle_tgt[lane];
        }
      }
      // (a launch that leaves records has its done flags written by the output epilogue, from slot 31; one that keeps
      // only the view's flag - done_stride = 0 - writes the LAST step's: in a migrating launch the steps of a robot run
      // on waves of different XCDs, whose L2s would write their plain stores to the one byte back in any order)
      if (B.traj == nullptr && lane == 0 && (B.done_stride != 0 || step == B.steps - 1)) B.done[(size_t)step * B.done_stride + env] = done ? 1 : 0;
    }
    SOLO_STAMP(B, 12);
    wave_sync();  // this step's LDS state is complete before the next step reads it
  }
  SOLO_STAMP(B, 13);
  const int lane1 = sizeof(T) == 8 ? wave_fresh_lane() : wave_opaque_lane(lane0);  // re-derive the addresses instead of keeping them live
  // ---- THE OUTPUT EPILOGUE (round 3): the launch's observations, rewards, done flags and episodic bookkeeping,
  //      evaluated by the robot's own wave from the records it left, 32 steps per pass with LANE = STEP - one pass
  //      costs what one item costs (~450 instructions), whatever the number of steps in it: 0.2 % of a 250-step
  //      launch, 3 % of a 20-step one, and a wave that finishes early does this while the launch waits for its slowest
  //      robot anyway.  Rounds 1-2 ran two more kernels after the launch (one thread per robot-step; 15 + 5 us and two
  //      launch gaps per 0.36-ms 20-step rollout in f32, 44 + 6 us in f64); the per-item functions are the same
  //      (solo_outputs.h), so are the results, bit for bit.  The records are re-read from global memory (this wave
  //      wrote them: L2-resident, its own robot's are contiguous); the reward program's values live in the row-vector
  //      block of LDS, which is dead by now; lane 0 then folds the pass's rewards into the episodic accumulators in
  //      step order (accumulate_returns: the additions stay sequential).
  //      With robot migration every chunk's wave does this for the steps of ITS chunk (it reads only records it wrote
  //      itself; the episodic accumulators travel in the robot's record) and goes on to its next task afterwards - so
  //      the scratch stays clear of the per-launch tables (28 steps per pass in f64).
  if constexpr (kFull) if (B.traj != nullptr) {
    wave_fence_global();  // this wave's record stores before its loads of them
    SOLO_STAMP_E(B, 1);
    const auto A = wave_cold_args(Bin);
    const int n_obs = wave_uniform(s_const.num_obs), n_rops = wave_uniform(s_const.num_reward_ops);
    constexpr int kPass = kRowsReals / SOLO_MAX_REWARD_OPS < 32 ? kRowsReals / SOLO_MAX_REWARD_OPS : 32;
    static_assert(kPass >= 16 && sizeof(T) * 32 >= (size_t)kPass, "the output epilogue's scratch");
    T* const val = s_blk;                                            // [n_rops][kPass]: the row vectors' block (dead here)
    uint8_t* const ev_bytes = reinterpret_cast<uint8_t*>(s_keep);    // (the parked factors are dead too)
    const T* const my_traj = B.traj + (size_t)(env - B.env_base) * (size_t)B.steps * SOLO_STATE_STRIDE;
    const bool want_reward = (B.flags & SOLO_STEP_REWARD) != 0;
    const bool bookkeeping = want_reward && (B.flags & SOLO_STEP_DONE) != 0;
    T* const obs_rec = A->obs_rec; T* const reward_rec = A->reward_rec;
    T* const view_obs = A->view_obs; T* const view_reward = A->view_reward; uint8_t* const view_done = A->view_done;
    const long long obs_stride = A->obs_rec_stride, reward_stride = A->reward_rec_stride;
    const int obs_from = A->obs_from;
    for (int base = step_begin; base < step_end; base += kPass) {
      const int k = base + lane1;
      if (lane1 < kPass && k < step_end) {
        const T* rec = my_traj + (size_t)k * SOLO_STATE_STRIDE;
        const bool last = k == B.steps - 1;
        const int ev = (int)rec[SOLO_S_SPARE];
        ev_bytes[lane1] = (uint8_t)ev;
        if (B.flags & SOLO_STEP_DONE) {
          if (B.done_stride != 0 || last) B.done[(size_t)k * B.done_stride + env] = (uint8_t)(ev & kEventDone);
          if (view_done != nullptr && last) view_done[env] = (uint8_t)(ev & kEventDone);
        }
        SOLO_STAMP_E(B, 2);
        T roll, pitch, yaw;
        euler_from_quat<T>(rec[SOLO_S_QUAT], rec[SOLO_S_QUAT + 1], rec[SOLO_S_QUAT + 2], rec[SOLO_S_QUAT + 3], &roll, &pitch, &yaw);
        SOLO_STAMP_E(B, 3);
        if (B.flags & SOLO_STEP_OBS) {
          T* o_rec = (obs_rec != nullptr && k >= obs_from) ? obs_rec + (size_t)k * obs_stride + (size_t)env * n_obs : nullptr;
          T* o_view = (view_obs != nullptr && last) ? view_obs + (size_t)env * n_obs : nullptr;
          if (o_rec != nullptr || o_view != nullptr)
            for (int i = 0; i < n_obs; ++i) {
              const T x = observation_value<T>(P0->obs[i], rec, roll, pitch, yaw);
              if (o_rec != nullptr) o_rec[i] = x;
              if (o_view != nullptr) o_view[i] = x;
            }
        }
        SOLO_STAMP_E(B, 4);
        if (want_reward) {
          const T rv = eval_reward<T>(P0, rec, roll, pitch, val + lane1, kPass);
          if (reward_rec != nullptr) reward_rec[(size_t)k * reward_stride + env] = rv;
          if (view_reward != nullptr && last) view_reward[env] = rv;
        }
        SOLO_STAMP_E(B, 5);
      }
      wave_sync();
      SOLO_STAMP_E(B, 6);
      if (bookkeeping && lane1 == 0) {
        const int cnt = step_end - base < kPass ? step_end - base : kPass;
        accumulate_returns<T>(s_state, ev_bytes, 1, val + (size_t)(n_rops - 1) * kPass, 1, cnt, SOLO_STATS_ROW,
                              [](double* p, double x) { stats_add(p, x); });
      }
      wave_sync();
      SOLO_STAMP_E(B, 7);
    }
  }
  if ((B.flags & SOLO_STEP_DONE) && lane1 < SOLO_MAX_TERMS) {
    if constexpr (kMigrate) wave_atomic_store(wave_cold_args(Bin)->term_count + (size_t)env * SOLO_MAX_TERMS + lane1, s_cnt[lane1]);
    else wave_cold_args(Bin)->term_count[(size_t)env * SOLO_MAX_TERMS + lane1] = s_cnt[lane1];
  }
  if ((B.flags & SOLO_STEP_PHYSICS) && lane1 == 0 && last_chunk) { int32_t* cost = wave_cold_args(Bin)->cost; if (cost != nullptr) cost[env] = prio_sweeps - hist_sweeps; }
  // (slots SOLO_S_RETURN.. of the record: the episodic accumulators, kept by whichever path evaluated the rewards)
  const bool own_returns = (B.flags & SOLO_STEP_REWARD) && (B.flags & SOLO_STEP_DONE) &&
                           (B.traj != nullptr || (kInlineOutputs<T, kFull> && B.reward_inline != nullptr));
  if (lane1 < (own_returns ? SOLO_S_SPARE : SOLO_S_RETURN)) {
    if constexpr (kMigrate) wave_store_shared(wave_cold_args(Bin)->state + rec + lane1, s_state[lane1]);
    else wave_cold_args(Bin)->state[rec + lane1] = s_state[lane1];
  }
  SOLO_STAMP(B, 14);
#if defined(SOLO_STAMPS) && !defined(SOLO_STAMPS_LIGHT)
  wave_sync();
  if (lane1 < 16) B.stamps[(size_t)env * 32 + 16 + lane1] = s_acc[lane1];
#endif
  if constexpr (kMigrate) {
    // hand the robot on (unless this was its last chunk): its history, then - when the device-coherent stores of its
    // record and counters above have completed - its number and next chunk into the next free slot of its ring (whoever
    // holds that slot's ticket continues it).  The slot's index and the wave's OWN next ticket are two independent
    // read-modify-writes: both are issued here, in front of the one wait that the stores need anyway - a hand-over is
    // a chain of device-scope round trips (~1.8 us each under load), and this takes two of the five off it.
    const int chunks = migration_chunks(B.steps, B.q_chunk), ring_len = (B.count / B.q_rings) * chunks;
    int at = 0, nt = 0;
    if (lane1 == 0) {
      if (!last_chunk) {
        wave_atomic_store(queue + kQueueHeader + (env - B.env_base), prio_sweeps);
        at = wave_atomic_add(queue + q_ring * 32 + 16, 1);
      }
      nt = wave_atomic_add(queue + q_ring * 32, 1);
    }
    wave_release_device();
    if (!last_chunk && lane1 == 0)
      wave_atomic_store(queue + kQueueHeader + (size_t)B.count + (size_t)q_ring * ring_len + at, (env - B.env_base) | ((q_chunk_at + 1) << 24));
    next_ticket = wave_readlane_int(nt, 0);
    have_next = true;
  }
  if constexpr (kMigrate) wave_sync();  // (the next task's prologue rewrites the LDS record)
  } while (kMigrate);  // the task loop
}

// the work queue of a launch with robot migration (solo_kernel_params.h): one thread per entry
static __global__ void solo_queue_init_kernel(int32_t* __restrict__ q, size_t ints, int env_base, int n, int rings, int steps, int chunk,
                                       const int32_t* __restrict__ order) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ints) migration_queue_init(q, i, env_base, n, rings, steps, chunk, order);
}

// setJointMotorControlArray without a step (solo8v2vanilla.py:87-90)
template <typename T>
__global__ void solo_set_targets_kernel(const T* __restrict__ actions, T* __restrict__ targets, T scale, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) targets[i] = actions[i] * scale;
}

// resetSimulation + settle, as a masked snapshot restore (solo8v2vanilla.py:104-143)
template <typename T>
__global__ void solo_reset_kernel(const KParams<T>* __restrict__ P, T* __restrict__ state, const T* __restrict__ snapshot,
                                  T* __restrict__ targets, int32_t* __restrict__ term_count, T* __restrict__ warm,
                                  const uint8_t* __restrict__ mask, int num_envs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int env = i / SOLO_STATE_STRIDE, e = i % SOLO_STATE_STRIDE;
  if (env >= num_envs) return;
  if (mask != nullptr && mask[env] == 0) return;
  state[i] = snapshot[i];
  if (e < SOLO_MAX_TERMS) term_count[env * SOLO_MAX_TERMS + e] = 0;
  warm[(size_t)env * 64 + 2 * e] = warm[(size_t)env * 64 + 2 * e + 1] = T(0);  // (a reset robot starts from zero impulses)
  // the settle loop ends with the motors commanded to the settle pose (solo8v2vanilla.py:127-136)
  if (e < SOLO_NUM_JOINTS) targets[env * SOLO_NUM_JOINTS + e] = P->c.settle_tgt[e];
}

// loadURDF at robot_start_pos / orientation (solo8v2vanilla.py:151-155), zero velocities
template <typename T>
__global__ void solo_init_kernel(T* __restrict__ state, T* __restrict__ targets, T px, T py, T pz, T qx, T qy, T qz, T qw, int num_envs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int env = i / SOLO_STATE_STRIDE, e = i % SOLO_STATE_STRIDE;
  if (env >= num_envs) return;
  T v = T(0);
  if (e == SOLO_S_POS) v = px; else if (e == SOLO_S_POS + 1) v = py; else if (e == SOLO_S_POS + 2) v = pz;
  else if (e == SOLO_S_QUAT) v = qx; else if (e == SOLO_S_QUAT + 1) v = qy; else if (e == SOLO_S_QUAT + 2) v = qz;
  else if (e == SOLO_S_QUAT + 3) v = qw;
  state[i] = v;
  if (e < SOLO_NUM_JOINTS) targets[env * SOLO_NUM_JOINTS + e] = T(0);
}

}  // namespace solo
