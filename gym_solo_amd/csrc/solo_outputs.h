// solo_outputs.h — observations, rewards and episodic returns, evaluated PER ROBOT-STEP from the
// state records the step kernel leaves behind.
//
// The step kernel (solo_step_kernel.h) is one wavefront per robot and bound by dependent-issue
// latency; the reductions of the reference's factories
//   ObservationFactory.get_obs   gym_solo/core/obs.py:130-159 (TorsoIMU :268-279, MotorEncoder :354-362)
//   RewardFactory.get_reward     gym_solo/core/rewards.py:104-118 (+ the reward classes :121-373)
// are pure functions of one robot's state after a step.  Evaluated inside every step of the robot's wave
// (lane = item) they were another ~25 % of serial latency per step (measured); a fused launch therefore
// leaves one record per robot-step and evaluates them AFTER its last step, one item per LANE (lane =
// step: the output epilogue of the step kernel; rounds 1-2: separate kernels, one item per thread).  The
// functions below are plain per-item code (no wave operations): the step kernel's epilogue calls them per
// lane, its single-step f32 path per item, and the CPU emulator (tests/emu) runs both.
//
// What the step kernel leaves behind per robot-step:
//   traj   [N][steps][SOLO_STATE_STRIDE] reals: [0..28] the state after the step (before an
//          auto-reset restores the snapshot), SOLO_S_* offsets; [31] the step's event bits:
//          1 = done, 2 = restarted from the snapshot
#pragma once

#include "solo_kernel_params.h"

namespace solo {

constexpr int kEventDone = 1, kEventRestart = 2;

// Euler angles of pybullet.getEulerFromQuaternion ([recalled] pybullet.c; call sites
// gym_solo/core/obs.py:271, rewards.py:233,265; known answer test_obs_observations.py:67-88):
//   |sarg| < 0.99999:  roll = atan2(2 (yz + wx), ww - xx - yy + zz), pitch = asin(sarg), yaw = atan2(2 (xy + wz), ww + xx - yy - zz)
//   sarg <= -0.99999:  roll = 0, pitch = -pi/2, yaw = 2 atan2(x, -y);   sarg >= 0.99999:  roll = 0, pitch = pi/2, yaw = 2 atan2(-x, y)
// Every angle is ONE atan2 and a fix-up - asin(s) = atan2(s, sqrt(1 - s^2)) - so that component `which` (0 roll, 1 pitch,
// 2 yaw) can be a LANE's job: the in-place outputs of a single-step launch evaluate the three angles on three lanes with
// one atan2 (round 6; one after the other they were three library calls at the end of every closed-loop f64 step), the
// output epilogue (lane = step) calls it three times - the same function with the same arguments, so the two paths agree
// bit for bit.
template <typename T>
__device__ __forceinline__ T euler_component(int which, T x, T y, T z, T w) {
  using R = Real<T>;
  // (EXPLICIT fused multiply-adds: this function is inlined with `which` a constant (the epilogue) and with `which` a lane's
  // number (the in-place outputs) - left to -ffp-contract the two copies were contracted differently and a fused rollout's
  // observations differed from single steps' in the last bit)
  const T sarg = T(-2) * R::fma(x, z, -(w * y));
  const bool lo = sarg <= T(-0.99999), hi = sarg >= T(0.99999);
  const T ww_zz = R::fma(w, w, z * z), xx_yy = R::fma(x, x, y * y), ww_xx = R::fma(w, w, x * x), yy_zz = R::fma(y, y, z * z);
  T ya, xa;
  if (which == 0) { ya = T(2) * R::fma(y, z, w * x); xa = ww_zz - xx_yy; }      // (ww - xx - yy + zz)
  else if (which == 1) { ya = sarg; xa = R::cos_of_asin(sarg); }
  else {
    ya = T(2) * R::fma(x, y, w * z); xa = ww_xx - yy_zz;                          // (ww + xx - yy - zz)
    if (lo) { ya = x; xa = -y; }
    if (hi) { ya = -x; xa = y; }
  }
  const T a = R::atan2(ya, xa);
  const T half_pi = R::half_pi();
  if (which == 0) return (lo || hi) ? T(0) : a;
  if (which == 1) return lo ? -half_pi : (hi ? half_pi : a);
  return (lo || hi) ? T(2) * a : a;
}
template <typename T>
__device__ __forceinline__ void euler_from_quat(T x, T y, T z, T w, T* roll, T* pitch, T* yaw) {
  *roll = euler_component<T>(0, x, y, z, w);
  *pitch = euler_component<T>(1, x, y, z, w);
  *yaw = euler_component<T>(2, x, y, z, w);
}

// gaussian tolerance, gym_solo/core/rewards.py:384-431 with margin_value = 0.1;
// scale_over_margin = sqrt(-2 ln 0.1) / margin is prepared on the host (0 when margin = 0)
template <typename T>
__device__ __forceinline__ T tolerance(T x, T lo, T hi, T margin, T scale_over_margin) {
  const bool within = (lo <= x) && (x <= hi);
  if (margin == T(0)) return within ? T(1) : T(0);
  const T t = ((x < lo) ? (lo - x) : (x - hi)) * scale_over_margin;
  const T v = Real<T>::exp(T(-0.5) * (t * t));
  return within ? T(1) : v;
}

// element `src` of the observation source vector (include/solo_engine.h, SOLO_SRC_*); `src` is the
// same for every item of a launch, so the branches are uniform
template <typename T>
__device__ __forceinline__ T source_value(const T* rec, int src, T roll, T pitch, T yaw) {
  if (src < 3) return src == 0 ? roll : (src == 1 ? pitch : yaw);
  if (src < 6) return rec[SOLO_S_LINVEL + src - 3];
  if (src < 9) return rec[SOLO_S_ANGVEL + src - 6];
  if (src < 33) {
    const int j = (src - 9) % 12, off = src < 21 ? SOLO_S_Q : SOLO_S_QD;
    return (j % 3 == 2) ? T(0) : rec[off + 2 * (j / 3) + (j % 3)];  // fixed ANKLE joints read 0
  }
  if (src < 36) return rec[SOLO_S_POS + src - 33];
  if (src < 40) return rec[SOLO_S_QUAT + src - 36];
  return T(1);
}

// one element of the observation program: source, scale, clip, normalise (obs.py:141-159)
template <typename T>
__device__ __forceinline__ T observation_value(const ObsElemK<T>& e, const T* rec, T roll, T pitch, T yaw) {
  using R = Real<T>;
  T v = source_value<T>(rec, e.src, roll, pitch, yaw) * e.scale;
  if (e.flags & 1) v = R::min(R::max(v, e.lo), e.hi);
  if (e.flags & 2) v = R::fma(v, e.nscale, e.noff);
  return v;
}
template <typename T>
__device__ __forceinline__ void eval_observations(const KParams<T>* P, const T* rec, T roll, T pitch, T yaw, T* out) {
  const int n = P->c.num_obs;
  for (int i = 0; i < n; ++i) out[i] = observation_value<T>(P->obs[i], rec, roll, pitch, yaw);
}

// Reward program in three-address form (pack_program): instruction i's value goes to
// val[i * stride]; LEAVES read the state, SCALE / ADD / MUL combine earlier values
// (rewards.py:104-118: the weighted sum is compiled into the program).
__device__ __forceinline__ bool reward_is_leaf(int op) { return op < SOLO_R_SCALE; }
// One leaf of the reward program.  The four tolerance leaves differ only in WHAT they measure: every case prepares
// (x or x^2, bounds, margin) and ONE sqrt and ONE tolerance() - the exp - follow for all of them.  In the in-place path of a
// single-step launch lane i evaluates instruction i, so the leaf types of a program sit on different lanes of one wave: as
// four cases with a tolerance() each they ran one after the other under EXEC masks (round 5: ~1.5 k cycles of a closed-loop
// step's 8.8 k of outputs); in the output epilogue (lane = step: the instruction is the same on every lane) nothing changes.
// Same arithmetic per leaf as before, bit for bit.
template <typename T>
__device__ __forceinline__ T reward_leaf(const RewardInstrK<T>& r, const T* rec, T roll, T pitch) {
  using R = Real<T>;
  T x = T(0), lo = T(0), hi = T(0), margin = T(0), direct = T(0);
  bool root = false, tol = true;
  switch (r.op) {
    case SOLO_R_UPRIGHT: {  // rewards.py:221-234: pitch relative to "fully upright" = -pi/2
      const T fu = T(-1.5707963267948966);
      direct = fu * pitch / (fu * fu);
      tol = false;
      break;
    }
    case SOLO_R_FLAT_TORSO:  // rewards.py:256-269: tolerance(sqrt(roll^2 + pitch^2), (-a, a), b)
      x = R::fma(roll, roll, pitch * pitch); root = true;   // (explicit: the same bits in every inlined copy)
      lo = -r.a; hi = r.a; margin = r.b;
      break;
    case SOLO_R_TORSO_HEIGHT:  // rewards.py:362-373: tolerance(z, (a - b, a + b), c)
      x = rec[SOLO_S_POS + 2];
      lo = r.a - r.b; hi = r.a + r.b; margin = r.c;
      break;
    case SOLO_R_HORIZ_SPEED: {  // rewards.py:326-338: tolerance(|v_xy|, (a - b, a + b), c)
      const T vx = rec[SOLO_S_LINVEL], vy = rec[SOLO_S_LINVEL + 1];
      x = R::fma(vx, vx, vy * vy); root = true;
      lo = r.a - r.b; hi = r.a + r.b; margin = r.c;
      break;
    }
    case SOLO_R_SMALL_CONTROL: {  // rewards.py:290-301: mean |joint rate| over all 12 joints
      T sum = T(0);
      for (int j = 0; j < SOLO_NUM_DOF; ++j) sum += R::abs(rec[SOLO_S_QD + j]);
      x = sum / T(SOLO_NUM_JOINTS);
      margin = r.a;
      break;
    }
    case SOLO_R_CONST: direct = r.a; tol = false; break;
    default: tol = false; break;
  }
  if (!tol) return direct;
  if (root) x = R::sqrt(x);
  return tolerance<T>(x, lo, hi, margin, r.d);
}
template <typename T>
__device__ __forceinline__ T reward_combine(const RewardInstrK<T>& r, const T* val, int stride) {
  const T x0 = val[(r.src & 255) * stride];
  if (r.op == SOLO_R_SCALE) return r.a * x0;
  const T x1 = val[((r.src >> 8) & 255) * stride];
  return r.op == SOLO_R_ADD ? x0 + x1 : x0 * x1;
}
// the whole program for one item; returns the value of the last instruction
template <typename T>
__device__ __forceinline__ T eval_reward(const KParams<T>* P, const T* rec, T roll, T pitch, T* val, int stride) {
  const int n = P->c.num_reward_ops;
  T last = T(0);
  for (int i = 0; i < n; ++i) {
    const RewardInstrK<T>& r = P->reward[i];
    last = reward_is_leaf(r.op) ? reward_leaf<T>(r, rec, roll, pitch) : reward_combine<T>(r, val, stride);
    val[i * stride] = last;
  }
  return last;
}

// One robot's episodic bookkeeping over the `steps` steps of a launch, in step order: the
// return / length accumulators live in the robot's state record (SOLO_S_RETURN / SOLO_S_EPLEN);
// a restart closes them (and, when the episode really ended, adds it to the statistics shard).
// The rewards and event bytes of a pass sit in LDS (the output epilogue) or are a single step's (the in-place
// path): a plain loop in step order - no local arrays, which the compiler indexes dynamically through
// s_set_gpr_idx_on (rounds 1-2 prefetched 16 steps at a time from global memory for the returns kernel).
template <typename T, typename AddFn>
__device__ __forceinline__ void accumulate_returns(T* state_rec, const uint8_t* events, long long events_stride, const T* reward,
                                                   long long reward_stride, int steps, double* stats, AddFn add) {
  T ret = state_rec[SOLO_S_RETURN], len = state_rec[SOLO_S_EPLEN];
#pragma unroll 1
  for (int k = 0; k < steps; ++k) {
    const int ev = events[(size_t)k * events_stride];
    ret += reward[(size_t)k * reward_stride];
    len += T(1);
    if (ev & kEventRestart) {
      if (ev & kEventDone) {
        const double x = (double)ret;
        add(&stats[0], x);
        add(&stats[1], x * x);
        add(&stats[2], 1.0);
        add(&stats[3], (double)len);
      }
      ret = T(0);
      len = T(0);
    }
  }
  state_rec[SOLO_S_RETURN] = ret;
  state_rec[SOLO_S_EPLEN] = len;
}

}  // namespace solo
