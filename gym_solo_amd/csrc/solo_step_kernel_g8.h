// solo_step_kernel_g8.h — EXPERIMENT (never the product library): the leg dynamics of EIGHT robots in ONE wave.
//
// The product kernel (solo_step_kernel.h) gives every robot a wavefront of its own; its dynamics phase -
// kinematics, composite inertias, Newton-Euler bias, the per-leg Schur terms, the 6x6 base Cholesky: ~650 of
// the ~1450 VALU instructions of an f32 env step - computes 8 distinct lanes' worth of numbers (4 legs x 2
// links) on 64 lanes, i.e. 8-way redundantly.  The round-2 review asked for a measurement instead of an
// argument: this file is the dynamics phase for the build `make group8` (-DSOLO_GROUP8), in which a workgroup
// is 8 waves serving 8 robots.  Per env step:
//     barrier | wave 0: this function - lane = 8 robot + 2 leg + link, the SAME arithmetic as the product's
//             | dynamics phase, per-leg sums over DPP quad permutes instead of LDS, results parked per robot in LDS
//     barrier | every wave: its own robot's constraint rows, Delassus columns, Gauss-Seidel, integration
// so the dynamics instructions are issued once per 8 robots - and 8 robots advance in lockstep, one barrier pair
// per step.  tests/test_gpu_group8.py compares the build with the product library BIT FOR BIT;
// tools/ab_group8.sh measures it (profiles/round3_group8_ab.log).
#pragma once

#include "solo_kernel_params.h"

namespace solo {

// lane = 8 r + 2 leg + link inside the dynamics wave
struct LaneMap8 {
  // x[lane ^ 1]: the other link of the leg (DPP quad_perm [1, 0, 3, 2])
  template <typename T> static __device__ __forceinline__ T other_link(T x) { return dpp_mov<0xB1>(x); }
  // the lower link's (link 1) / the upper link's value in both lanes of the leg (quad_perm [1, 1, 3, 3] / [0, 0, 2, 2])
  template <typename T> static __device__ __forceinline__ T from_lower(T x) { return dpp_mov<0xF5>(x); }
  template <typename T> static __device__ __forceinline__ T from_upper(T x) { return dpp_mov<0xA0>(x); }
  // sum over the four legs of a robot of a value that is identical in both link lanes of each leg:
  // (leg 0 + leg 1) + (leg 2 + leg 3), the association of the product kernel's LDS sum
  // (quad_perm [2, 3, 0, 1] = lane ^ 2, then row_half_mirror: lane -> 7 - lane inside the robot's 8 lanes)
  template <typename T> static __device__ __forceinline__ T sum_legs(T x) {
    x = x + dpp_mov<0x4E>(x);
    return x + dpp_mov<0x141>(x);
  }
};

template <typename T> __device__ __forceinline__ T g8_both(T x) { return x + LaneMap8::other_link(x); }
template <typename T> __device__ __forceinline__ V3<T> g8_both(V3<T> v) { return {g8_both(v.x), g8_both(v.y), g8_both(v.z)}; }
template <typename T> __device__ __forceinline__ T g8_lower(T x) { return LaneMap8::from_lower(x); }
template <typename T> __device__ __forceinline__ V3<T> g8_lower(V3<T> v) { return {g8_lower(v.x), g8_lower(v.y), g8_lower(v.z)}; }

constexpr int kG8 = 8;          // robots per workgroup
constexpr int kG8LegSlots = 24;  // s_leg row: the product's 19 values + cos / sin of the two link angles

// The dynamics phase of physics_solve (solo_step_kernel.h: from "base: rotation" to "Park the factors"), for the
// 8 robots of a workgroup at once.  Expression by expression the product's code; only the lane mapping differs.
// s_state [8][32], s_keep [8][32], s_leg [8][4][kG8LegSlots], s_mass [8] (base-mass scale of each robot).
template <typename T>
__device__ __forceinline__ void physics_dynamics_g8(const StepConst<T>& C, const LegConst<T>* s_legc, const T* s_state_g, T* s_keep_g,
                                                    T* s_leg_g, const T* s_mass_g, const T* s_math, int lane) {
  using R = Real<T>;
  const int robot = lane >> 3, leg = (lane >> 1) & 3;
  const bool lower = (lane & 1) != 0;
  const T* s_state = s_state_g + robot * SOLO_STATE_STRIDE;
  const LegConst<T>& L = s_legc[leg];
  const T mass_scale = s_mass_g[robot];
  const T dt = C.dt;

  const T qx = s_state[SOLO_S_QUAT], qy = s_state[SOLO_S_QUAT + 1], qz = s_state[SOLO_S_QUAT + 2], qw = s_state[SOLO_S_QUAT + 3];
  const T r00 = T(1) - T(2) * (qy * qy + qz * qz), r01 = T(2) * (qx * qy - qw * qz), r02 = T(2) * (qx * qz + qw * qy);
  const T r10 = T(2) * (qx * qy + qw * qz), r11 = T(1) - T(2) * (qx * qx + qz * qz), r12 = T(2) * (qy * qz - qw * qx);
  const T r20 = T(2) * (qx * qz - qw * qy), r21 = T(2) * (qy * qz + qw * qx), r22 = T(1) - T(2) * (qx * qx + qy * qy);
  const V3<T> ww = {s_state[SOLO_S_ANGVEL], s_state[SOLO_S_ANGVEL + 1], s_state[SOLO_S_ANGVEL + 2]};
  const V3<T> vw = {s_state[SOLO_S_LINVEL], s_state[SOLO_S_LINVEL + 1], s_state[SOLO_S_LINVEL + 2]};
  const V3<T> gw = {C.gravity[0], C.gravity[1], C.gravity[2]};
  const V3<T> om = {r00 * ww.x + r10 * ww.y + r20 * ww.z, r01 * ww.x + r11 * ww.y + r21 * ww.z, r02 * ww.x + r12 * ww.y + r22 * ww.z};
  const V3<T> vb = {r00 * vw.x + r10 * vw.y + r20 * vw.z, r01 * vw.x + r11 * vw.y + r21 * vw.z, r02 * vw.x + r12 * vw.y + r22 * vw.z};
  const V3<T> gb = {r00 * gw.x + r10 * gw.y + r20 * gw.z, r01 * gw.x + r11 * gw.y + r21 * gw.z, r02 * gw.x + r12 * gw.y + r22 * gw.z};

  const T bm = lower ? T(1) : T(0);
  const T q1 = s_state[SOLO_S_Q + 2 * leg], q2 = s_state[SOLO_S_Q + 2 * leg + 1];
  const T qd1 = s_state[SOLO_S_QD + 2 * leg], qd2 = s_state[SOLO_S_QD + 2 * leg + 1];
  T sinb, cosb;
  R::sincos(lower ? q1 + q2 : q1, &sinb, &cosb, s_math);
  const T s1 = LaneMap8::from_upper(sinb), c1 = LaneMap8::from_upper(cosb);
  const T s12 = LaneMap8::from_lower(sinb), c12 = LaneMap8::from_lower(cosb);
  const V3<T> o1 = {L.hip[0], L.hip[1], L.hip[2]};
  const V3<T> o2 = o1 + roty(c1, s1, V3<T>{L.knee[0], L.knee[1], L.knee[2]});
  const V3<T> ob = select(lower, o2, o1);
  const T* body = L.link[lower ? 1 : 0];
  const T mB = body[0];
  const V3<T> c = ob + roty(cosb, sinb, V3<T>{body[1], body[2], body[3]});
  T I[6];
  rot_inertia_y(cosb, sinb, body + 4, I);

  const V3<T> r1 = c - o1, r2 = c - o2;
  const V3<T> t1 = ycross(r1), t2 = ycross(r2);
  const V3<T> Iy = {I[3], I[1], I[5]};
  const V3<T> f1 = g8_both(mB * t1);
  const V3<T> n1 = g8_both(Iy + mB * cross(c, t1));
  const T P11 = g8_both(mB * dot(t1, t1) + I[1]);
  const V3<T> f2 = g8_lower(mB * t2);
  const V3<T> n2 = g8_lower(Iy + mB * cross(c, t2));
  const T P12 = g8_lower(mB * dot(t1, t2) + I[1]);
  const T P22 = g8_lower(mB * dot(t2, t2) + I[1]);
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

  const V3<T> wU = {om.x, om.y + qd1, om.z};
  const V3<T> wB = {om.x, wU.y + bm * qd2, om.z};
  const V3<T> aU = {-qd1 * om.z, T(0), qd1 * om.x};
  const V3<T> a = {aU.x - (bm * qd2) * wU.z, T(0), aU.z + (bm * qd2) * wU.x};
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
  const T h1 = g8_both(N.y + (r1.z * F.x - r1.x * F.z));
  const T h2 = g8_lower(N.y + (r2.z * F.x - r2.x * F.z));
  const V3<T> Fleg = g8_both(F);
  const V3<T> Nleg = g8_both(N + cross(c, F));
  const T e1 = h1 * iL11, e2 = (h2 - L21 * e1) * iL22;
  const T y2 = e2 * iL22, y1 = (e1 - L21 * y2) * iL11;

  const T mleg = L.link[0][0] + L.link[1][0];
  const V3<T> mc = g8_both(mB * c);
  T IO[6];
  IO[0] = g8_both(I[0] + mB * (c.y * c.y + c.z * c.z));
  IO[1] = g8_both(I[1] + mB * (c.x * c.x + c.z * c.z));
  IO[2] = g8_both(I[2] + mB * (c.x * c.x + c.y * c.y));
  IO[3] = g8_both(I[3] - mB * c.x * c.y);
  IO[4] = g8_both(I[4] - mB * c.x * c.z);
  IO[5] = g8_both(I[5] - mB * c.y * c.z);
  T S[6][6];
  S[0][0] = IO[0]; S[1][0] = IO[3]; S[1][1] = IO[1]; S[2][0] = IO[4]; S[2][1] = IO[5]; S[2][2] = IO[2];
  S[3][0] = T(0);  S[3][1] = mc.z;  S[3][2] = -mc.y;
  S[4][0] = -mc.z; S[4][1] = T(0);  S[4][2] = mc.x;
  S[5][0] = mc.y;  S[5][1] = -mc.x; S[5][2] = T(0);
  S[3][3] = mleg; S[4][3] = T(0); S[4][4] = mleg; S[5][3] = T(0); S[5][4] = T(0); S[5][5] = mleg;
  T rhs[6] = {-Nleg.x, -Nleg.y, -Nleg.z, -Fleg.x, -Fleg.y, -Fleg.z};
  // the 27 per-leg terms summed over the four legs of each robot: two DPP stages each (no LDS, no sync)
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) S[i][j] = LaneMap8::sum_legs(S[i][j] - W1[i] * W1[j] - W2[i] * W2[j]);
    rhs[i] = LaneMap8::sum_legs(rhs[i] + W1[i] * e1 + W2[i] * e2);
  }
  {
    const T m0 = C.base_mass * mass_scale;
    T I0[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) I0[i] = C.base_I[i] * mass_scale;
    S[0][0] += I0[0]; S[1][0] += I0[3]; S[1][1] += I0[1]; S[2][0] += I0[4]; S[2][1] += I0[5]; S[2][2] += I0[2];
    S[3][3] += m0; S[4][4] += m0; S[5][5] += m0;
    const V3<T> Iw0 = symmul(I0, om);
    const V3<T> N0 = cross(om, Iw0) + (ka * (T(1) + R::sqrt(om2))) * Iw0;
    const V3<T> F0 = m0 * ((kl * (T(1) + R::sqrt(dot(vb, vb)))) * vb - gb);
    rhs[0] -= N0.x; rhs[1] -= N0.y; rhs[2] -= N0.z;
    rhs[3] -= F0.x; rhs[4] -= F0.y; rhs[5] -= F0.z;
  }
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
  for (int i = 0; i < 6; ++i) { kx1 += K1[i] * xb[i]; kx2 += K2[i] * xb[i]; }
  const T ub[6] = {om.x + dt * xb[0], om.y + dt * xb[1], om.z + dt * xb[2],
                   vb.x + dt * xb[3], vb.y + dt * xb[4], vb.z + dt * xb[5]};
  const T us1 = qd1 + dt * (-y1 - kx1), us2 = qd2 + dt * (-y2 - kx2);

  // park per robot: the first lane of a robot the base factors, the upper-link lane of each leg the leg's
  T* s_keep = s_keep_g + robot * 32;
  if ((lane & 7) == 0) {
    int o = 0;
#pragma unroll
    for (int i = 1; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < i; ++j) s_keep[o++] = S[i][j];
#pragma unroll
    for (int i = 0; i < 6; ++i) { s_keep[15 + i] = iC[i]; s_keep[21 + i] = ub[i]; }
  }
  if (!lower) {
    T* sl = s_leg_g + (robot * 4 + leg) * kG8LegSlots;
#pragma unroll
    for (int i = 0; i < 6; ++i) { sl[i] = K1[i]; sl[6 + i] = K2[i]; }
    sl[12] = iL11; sl[13] = L21; sl[14] = iL22;
    sl[15] = us1; sl[16] = us2; sl[17] = q1; sl[18] = q2;
    sl[19] = c1; sl[20] = s1; sl[21] = c12; sl[22] = s12;
  }
}

}  // namespace solo
