// solo_wave_ops.h — wavefront-level primitives used by solo_step_kernel.h, gfx950 version.
// One workgroup == one 64-lane wavefront == one robot, so "block" and "wave" coincide and the
// LDS hand-offs between lanes need no s_barrier: DS operations of one wave execute in order.
// (tests/emu/wave_emu.h provides the same names for the CPU fibre emulator used by the
// sanitizer/parity tests; this file is the only one the product build includes.)
#pragma once

#include <hip/hip_runtime.h>

namespace solo {

__device__ __forceinline__ int lane_id() { return threadIdx.x; }
__device__ __forceinline__ int block_id() { return blockIdx.x; }

// Orders this wave's LDS writes before the following LDS reads of other lanes.
__device__ __forceinline__ void wave_sync() { __syncthreads(); }
// orders this wave's global stores before its later global loads of the same lines by OTHER lanes of the wave
// (the output epilogue re-reads the step records): release + acquire at workgroup scope = the waits, no cache operation
__device__ __forceinline__ void wave_fence_global() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// wave-uniform pointer made opaque to the optimiser (no instruction): loads through the result
// are not hoisted above this point
template <typename P> __device__ __forceinline__ P* wave_opaque(P* p) {
  asm volatile("" : "+s"(p));
  return p;
}
__device__ __forceinline__ unsigned long long wave_opaque_bits(unsigned long long x) {  // (wave-uniform bits)
  asm volatile("" : "+s"(x));
  return x;
}
__device__ __forceinline__ int wave_opaque_lane(int lane) {
  asm volatile("" : "+v"(lane));
  return lane;
}
// the lane number, computed HERE (v_mbcnt over an all-ones mask: two instructions, any EXEC): the step loop takes its
// lane from this at the top of every step.  Derived from the kernel's threadIdx register the lane was one more value
// live across the whole loop - and the f64 kernel, which lives on exactly 168 VGPRs, SPILLED it: a scratch reload per
// step whose s_waitcnt vmcnt(0) also waited for the step's freshly issued action load (round 4).
__device__ __forceinline__ int wave_fresh_lane() {
  int lane;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
  return lane;
}
__device__ __forceinline__ int wave_readlane_int(int x, int lane) { return __builtin_amdgcn_readlane(x, lane); }
// declares an int wave-uniform (v_readfirstlane -> SGPR)
__device__ __forceinline__ int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

__device__ __forceinline__ float wave_readlane(float x, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane));
}
__device__ __forceinline__ double wave_readlane(double x, int lane) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// ---- cross-lane sums without LDS traffic ---------------------------------------------------
// v_permlane16_swap / v_permlane32_swap (new on gfx950) exchange odd<->even 16-lane rows and
// the two 32-lane halves inside the VALU; DPP row rotations fold into the v_add itself.
// x[lane] + x[lane^16] + x[lane^32] + x[lane^48]: sum over the four 16-lane leg groups
__device__ __forceinline__ float wave_sum_legs(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ double wave_sum_legs(double x) {
  unsigned lo = (unsigned)(__double_as_longlong(x) & 0xffffffffll), hi = (unsigned)(__double_as_longlong(x) >> 32);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double s = __longlong_as_double(((long long)a[0] & 0xffffffffll) | ((long long)b[0] << 32)) +
                   __longlong_as_double(((long long)a[1] & 0xffffffffll) | ((long long)b[1] << 32));
  lo = (unsigned)(__double_as_longlong(s) & 0xffffffffll); hi = (unsigned)(__double_as_longlong(s) >> 32);
  auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __longlong_as_double(((long long)c[0] & 0xffffffffll) | ((long long)d[0] << 32)) +
         __longlong_as_double(((long long)c[1] & 0xffffffffll) | ((long long)d[1] << 32));
}
template <int CTRL> __device__ __forceinline__ float dpp_mov(float x) {
  // (bound_ctrl: lanes without a source read 0 - what `old` = 0 gave - without a zeroed register per move)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ double dpp_mov(double x) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// the value of lane ^ 8: the other half of the caller's 16-lane row (DPP row_ror:8)
template <typename T> __device__ __forceinline__ T wave_other_half16(T x) { return dpp_mov<0x128>(x); }
// the LOWER half's (lanes 8..15 of the row) value in both halves of each 16-lane row, or the UPPER half's:
// ONE bank-masked DPP move in place (row_ror:8 written only to the banks of the other half), instead of
// a zeroed temporary, a DPP move and a select
template <int BANKS> __device__ __forceinline__ float dpp_half_bcast(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x128, 0xf, BANKS, false));
}
template <int BANKS> __device__ __forceinline__ double dpp_half_bcast(double x) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp((int)(b & 0xffffffffll), (int)(b & 0xffffffffll), 0x128, 0xf, BANKS, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(b >> 32), (int)(b >> 32), 0x128, 0xf, BANKS, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <typename T> __device__ __forceinline__ T wave_from_lower_half16(T x) { return dpp_half_bcast<0x3>(x); }
template <typename T> __device__ __forceinline__ T wave_from_upper_half16(T x) { return dpp_half_bcast<0xc>(x); }
// the value of lane - N inside the caller's 16-lane row (DPP row_shr:N; the first N lanes of a row get 0)
template <int N, typename T> __device__ __forceinline__ T wave_lane_below(T x) { return dpp_mov<0x110 + N>(x); }
// the value of lane - N of the WHOLE wave (lanes < N get 0): DPP wave_shr:1 (a GFX9 control, like row_bcast), N times.
// The slot-space solver (ColumnBank<T>::kCompact) keeps a contact's rows in three consecutive SLOTS, which may straddle
// a 16-lane row: row_shr does not cross it.
template <int N, typename T> __device__ __forceinline__ T wave_slot_below(T x) {
#pragma unroll
  for (int i = 0; i < N; ++i) x = dpp_mov<0x138>(x);
  return x;
}
// number of set bits of a wave-uniform lane mask BELOW this lane (v_mbcnt_lo / v_mbcnt_hi)
__device__ __forceinline__ int wave_count_below(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
// PUSH: lane l's value goes to lane dst[l] (dst: a permutation of 0..63) - ds_permute_b32: the LDS crossbar, no LDS
// memory.  PULL: lane l gets the value of lane src[l] - ds_bpermute_b32.
__device__ __forceinline__ int wave_push_int(int x, int dst) { return __builtin_amdgcn_ds_permute(dst << 2, x); }
__device__ __forceinline__ float wave_push(float x, int dst) { return __int_as_float(__builtin_amdgcn_ds_permute(dst << 2, __float_as_int(x))); }
__device__ __forceinline__ double wave_push(double x, int dst) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_ds_permute(dst << 2, (int)(b & 0xffffffffll));
  const int hi = __builtin_amdgcn_ds_permute(dst << 2, (int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float wave_pull(float x, int src) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(x))); }
__device__ __forceinline__ double wave_pull(double x, int src) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_ds_bpermute(src << 2, (int)(b & 0xffffffffll));
  const int hi = __builtin_amdgcn_ds_bpermute(src << 2, (int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// sum over the 16 lanes of the caller's row, result in every lane (row_ror 8,4,2,1 all-reduce)
template <typename T> __device__ __forceinline__ T wave_sum_group16(T x) {
  x += dpp_mov<0x128>(x);
  x += dpp_mov<0x124>(x);
  x += dpp_mov<0x122>(x);
  x += dpp_mov<0x121>(x);
  return x;
}
// sum over all 64 lanes as a wave-uniform value: row all-reduce, then the GFX9 DPP row broadcasts
// (row_bcast:15 into rows 1,3; row_bcast:31 into rows 2,3) leave the total in lane 63, which
// v_readlane moves to a scalar register.  Association: (r3 + r2) + (r1 + r0) over the row sums.
__device__ __forceinline__ float wave_sum_all(float x) {
  x = wave_sum_group16(x);
  // written as DPP adds on the destination itself: rows outside row_mask keep their value (the
  // compiler cannot fold a masked mov_dpp + add for floats); s_nop covers the VALU-write -> DPP-read hazard
  asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(x));
  return wave_readlane(x, 63);
}
__device__ __forceinline__ double wave_sum_all(double x) { return wave_sum_legs(wave_sum_group16(x)); }
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __ballot(p); }

// Six wave-wide sums (z, returned wave-uniform) and two 16-lane sums (y, in every lane of the row) at
// once.  One at a time each reduction is a serial chain of DPP adds with an s_nop between every two
// stages (VALU write -> DPP read of the same register needs two wait states): ~16 instructions per
// sum, 114 for the eight.  Interleaved stage by stage the other seven chains fill the wait states:
// 44 DPP adds + 6 v_readlane, no nops, no dependent-issue stalls.  Same association as wave_sum_all /
// wave_sum_group16: row all-reduce (ror 8, 4, 2, 1), then row_bcast:15 into rows 1, 3 and
// row_bcast:31 into rows 2, 3, total in lane 63.
__device__ __forceinline__ void wave_reduce_rows(float (&z)[6], float (&y)[2]) {
#define SOLO_DPP_STAGE8(CTRL)                                                                      \
  "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                \
  "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                \
  "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                \
  "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                \
  "v_add_f32_dpp %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                \
  "v_add_f32_dpp %5, %5, %5 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                \
  "v_add_f32_dpp %6, %6, %6 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                \
  "v_add_f32_dpp %7, %7, %7 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
#define SOLO_DPP_STAGE6(CTRL, MASK)                                                                \
  "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"                           \
  "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"                           \
  "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"                           \
  "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"                           \
  "v_add_f32_dpp %4, %4, %4 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"                           \
  "v_add_f32_dpp %5, %5, %5 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"
  asm("s_nop 1\n\t"  // (the eight inputs were just written by VALU instructions)
      SOLO_DPP_STAGE8("row_ror:8") SOLO_DPP_STAGE8("row_ror:4") SOLO_DPP_STAGE8("row_ror:2") SOLO_DPP_STAGE8("row_ror:1")
      SOLO_DPP_STAGE6("row_bcast:15", "0xa") SOLO_DPP_STAGE6("row_bcast:31", "0xc")
      : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(y[0]), "+v"(y[1]));
#undef SOLO_DPP_STAGE8
#undef SOLO_DPP_STAGE6
#pragma unroll
  for (int i = 0; i < 6; ++i) z[i] = wave_readlane(z[i], 63);
}
__device__ __forceinline__ void wave_reduce_rows(double (&z)[6], double (&y)[2]) {
#pragma unroll
  for (int i = 0; i < 6; ++i) z[i] = wave_sum_all(z[i]);
  y[0] = wave_sum_group16(y[0]);
  y[1] = wave_sum_group16(y[1]);
}
// The same eight sums THROUGH LDS (f64, round 4).  As DPP chains a 64-bit sum costs three instructions per stage (two
// 32-bit DPP moves and the add: DPP takes no 64-bit operands) - ~156 for the eight.  Here every lane parks its eight
// terms (scratch[lane][9]: the odd stride keeps the column reads below free of bank conflicts), lane (k = lane & 7,
// part = lane >> 3) adds the terms k of the lanes 8 part ... 8 part + 7 in lane order, one row rotation joins the two
// parts of a 16-lane row (for k = 6, 7 that is the leg's sum of y: broadcast from the row's lanes 6 / 7 with
// row_newbcast), the permlane swaps join the four rows, and six v_readlane pairs hand out the totals: ~45 VALU
// instructions and two LDS round trips.  The caller orders its own LDS reads of the scratch area in front of this.
// ASSOCIATION (tests/emu/wave_emu.h restates it): ((((((x0 + x1) + x2) + x3) + x4) + x5) + x6) + x7 per part; part 2r +
// part 2r + 1 per row r; (row 0 + row 1) + (row 2 + row 3).
constexpr int kReduceScratch = 64 * 9;
template <int N> __device__ __forceinline__ double dpp_row_bcast(double x) {   // lane N of the caller's 16-lane row
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0x150 + N, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x150 + N, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ void wave_reduce_rows_lds(double (&z)[6], double (&y)[2], double* scratch, int lane) {
  double* const mine = scratch + lane * 9;
#pragma unroll
  for (int i = 0; i < 6; ++i) mine[i] = z[i];
  mine[6] = y[0]; mine[7] = y[1];
  wave_sync();
  const double* const col = scratch + (lane >> 3) * 72 + (lane & 7);
  double p = col[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) p += col[9 * i];
  p += dpp_mov<0x128>(p);
  const double total = wave_sum_legs(p);
#pragma unroll
  for (int i = 0; i < 6; ++i) z[i] = wave_readlane(total, i);
  y[0] = dpp_row_bcast<6>(p);
  y[1] = dpp_row_bcast<7>(p);
}

// ghat_s . ghat_r + hhat_s . x for the Delassus columns: the lane's own whitened row stays in
// registers, the other row comes from LDS.  f32: four v_pk_fma_f32 (two terms each) + one add.
typedef float solo_f32x2 __attribute__((ext_vector_type(2)));
template <typename T> struct RowDot {
  T g[6], h[2];
  __device__ __forceinline__ void set(const T* gh, const T* hh) {
#pragma unroll
    for (int i = 0; i < 6; ++i) g[i] = gh[i];
    h[0] = hh[0]; h[1] = hh[1];
  }
  // (explicit fused multiply-adds: the resident columns and the columns the overflow path of the slot-space solver
  // evaluates on the fly are the same bits in every inlined copy, whatever -ffp-contract decides elsewhere)
  __device__ __forceinline__ T dot(const T* rg, const T* rh) const {
    const T a1 = __builtin_fma(g[2], rg[2], __builtin_fma(g[1], rg[1], g[0] * rg[0]));
    const T a2 = __builtin_fma(g[5], rg[5], __builtin_fma(g[4], rg[4], g[3] * rg[3]));
    return (a1 + a2) + __builtin_fma(h[1], rh[1], h[0] * rh[0]);
  }
  // the same with the joint-space part counted only between rows of ONE leg: `same` = 1 (same leg) or 0 - one fused
  // multiply-add rounds exactly like the sum above (x 1) or leaves the base part alone (x 0)
  __device__ __forceinline__ T dot(const T* rg, const T* rh, T same) const {
    const T a1 = __builtin_fma(g[2], rg[2], __builtin_fma(g[1], rg[1], g[0] * rg[0]));
    const T a2 = __builtin_fma(g[5], rg[5], __builtin_fma(g[4], rg[4], g[3] * rg[3]));
    return __builtin_fma(same, __builtin_fma(h[1], rh[1], h[0] * rh[0]), a1 + a2);
  }
};
template <> struct RowDot<float> {
  solo_f32x2 g01, g23, g45, h01;
  __device__ __forceinline__ void set(const float* gh, const float* hh) {
    g01 = solo_f32x2{gh[0], gh[1]}; g23 = solo_f32x2{gh[2], gh[3]}; g45 = solo_f32x2{gh[4], gh[5]};
    h01 = solo_f32x2{hh[0], hh[1]};
  }
  __device__ __forceinline__ float dot(const float* rg, const float* rh) const {
    const solo_f32x2* r = reinterpret_cast<const solo_f32x2*>(rg);
    solo_f32x2 acc = g01 * r[0];
    acc = g23 * r[1] + acc;
    acc = g45 * r[2] + acc;
    acc = h01 * *reinterpret_cast<const solo_f32x2*>(rh) + acc;
    return acc.x + acc.y;
  }
};

// The scaled Delassus matrix of a robot.  Column r as lane s sees it:
//   col_r[s] = -(ghat_s . ghat_r + [same leg] hhat_s . hhat_r) / A_ss,  0 on the row's own lane.
// f32: one entry per (lane, column) RESIDENT IN REGISTERS - 64 column slots per lane held in two
// vector-typed register tuples (the widest the ISA has: 1024 bits), so that a wave-uniform column
// index turns into one register-indexed move (s_set_gpr_idx_on / v_mov_b32 / s_set_gpr_idx_off);
// a plain array would be demoted to scratch memory.  build() is only ever called with indices that
// are constants after unrolling; get() with a constant bank and a uniform r inside that bank.
typedef float solo_f32x32 __attribute__((ext_vector_type(32)));
template <typename T> struct ColumnBank;
template <> struct ColumnBank<float> {
  static constexpr bool kResident = true;
  static constexpr bool kCompact = false;    // lane = constraint row (fixed layout: solo_kernel_params.h)
  static constexpr int kSlots = 64, kRowStride = 8;
  static constexpr int kBanks = 2;
  static __device__ __forceinline__ constexpr unsigned long long bank_lanes(int b) { return 0xffffffffull << (32 * b); }
  RowDot<float> own;
  float nid;
  int lane;
  const float* rowvec;
  const float* hext;
  solo_f32x32 a0, a1;
  __device__ __forceinline__ void init(const float* gh, const float* hh, float nid_, int lane_, const float* rowvec_, const float* hext_) {
    own.set(gh, hh); nid = nid_; lane = lane_; rowvec = rowvec_; hext = hext_;
  }
  // (the own-lane zero is a select on the SCALE, not on the product: selecting the product makes the
  // compiler wrap every column's LDS reads in a divergent branch, one exposed round trip per column)
  __device__ __forceinline__ float column(int r) const { const float m = (lane == r) ? 0.0f : nid; return m * own.dot(rowvec + 8 * r, hext + 8 * r); }
  __device__ __forceinline__ void build(int r) { const float x = column(r); if (r < 32) a0[r] = x; else a1[r - 32] = x; }
  __device__ __forceinline__ float get(int bank, int r) const { return bank == 0 ? a0[r & 31] : a1[r & 31]; }
};
// f64 (the reference's precision): resident as well, but only for the LIVE rows (round 4).  64 doubles per lane were
// 128 VGPRs - half of a wave's registers at two waves per SIMD, and the reason for two waves per SIMD.  The rows that
// can move in a step are the 8 motor rows, three per sphere within the contact margin and the rare joint-limit rows:
// 8 + 3 x touching spheres - at most 32 in 99.97 % of the robot-steps of the benchmark workload, at most 44 on a
// plane (profiles/round4_live_rows_histogram.log).  So the f64 solver runs in SLOT space (kCompact): the step kernel
// permutes the live rows, in lane order, to the lanes 0 .. L-1 (solo_step_kernel.h), lane = slot holds slot's row,
// and the bank holds the columns of slots 0 .. 31 only: 32 doubles per lane = 64 VGPRs -> the kernel fits the 168
// VGPRs of THREE waves per SIMD.  A step with more than 32 live rows (a robot lying on everything it has) takes the
// overflow path: the same iteration with every column evaluated from LDS when it is used (column(), the expression
// build() stores) - slower per row, the same bits.  Row vectors in LDS: 8 doubles per slot (ghat 6, hhat 2) and the
// leg of the slot's row as a tag of its own (round 5).
typedef double solo_f64x16 __attribute__((ext_vector_type(16)));
template <> struct ColumnBank<double> {
  static constexpr bool kResident = true;
  static constexpr bool kCompact = true;
  static constexpr int kSlots = 32, kRowStride = 8;
  static constexpr int kBanks = 2;
  static __device__ __forceinline__ constexpr unsigned long long bank_lanes(int b) { return 0xffffull << (16 * b); }
  RowDot<double> own;
  double nid;
  int lane;
  const double* rowvec;        // [64 slots][ghat 6, hhat 2]
  const unsigned char* rowleg;  // [64 slots]: the leg of the slot's row ("same leg" is a compare of two tags: round 5 - rounds 3-4
  int leg;                      // kept a [64][4 legs x 2] array with the pair in the slot of the row's leg, 4 KB of LDS)
  solo_f64x16 a0, a1;
  __device__ __forceinline__ void init(const double* gh, const double* hh, double nid_, int lane_, const double* rowvec_, const unsigned char* rowleg_, int leg_) {
    own.set(gh, hh); nid = nid_; lane = lane_; rowvec = rowvec_; rowleg = rowleg_; leg = leg_;
  }
  __device__ __forceinline__ double same_leg(int leg_r) const { return leg_r == leg ? 1.0 : 0.0; }
  __device__ __forceinline__ double column(int r) const {
    const double m = (lane == r) ? 0.0 : nid;
    return m * own.dot(rowvec + kRowStride * r, rowvec + kRowStride * r + 6, same_leg(rowleg[r]));
  }
  __device__ __forceinline__ void build(int r) {
    const double x = column(r);
    if (r < 16) a0[r] = x; else a1[r - 16] = x;
  }
  // the same in two halves, for a build that fetches slot r + 1's row while it computes slot r's column (the LDS
  // broadcasts of the next row are in flight behind the arithmetic of this one: solo_step_kernel.h)
  struct Row { double g[6], h[2]; int leg; };
  __device__ __forceinline__ Row fetch(int r) const {
    Row x;
#pragma unroll
    for (int i = 0; i < 6; ++i) x.g[i] = rowvec[kRowStride * r + i];
    x.h[0] = rowvec[kRowStride * r + 6]; x.h[1] = rowvec[kRowStride * r + 7];
    x.leg = rowleg[r];
    return x;
  }
  __device__ __forceinline__ void build_from(int r, const Row& x) {
    const double m = (lane == r) ? 0.0 : nid;
    const double c = m * own.dot(x.g, x.h, same_leg(x.leg));
    if (r < 16) a0[r] = c; else a1[r - 16] = c;
  }
  __device__ __forceinline__ double get(int bank, int r) const { return bank == 0 ? a0[r & 15] : a1[r & 15]; }
};

// The kernel's by-value buffer block, re-read from the kernarg segment (constant address space ->
// s_load at the use site) instead of living in scalar registers for the whole launch: for the fields
// the step loop touches rarely or only in the prologue / epilogue (snapshot, statistics, targets,
// counters, cost).  Kept in registers they pushed the kernel over its SGPR budget (15 spills with
// reloads all over the step loop).  `by_value` is the kernel parameter itself (second parameter,
// after one 8-byte pointer).
template <typename B> using ColdArgs = const __attribute__((address_space(4))) B*;
template <typename B> __device__ __forceinline__ ColdArgs<B> wave_cold_args(const B& by_value) {
  (void)by_value;
  auto base = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  ColdArgs<B> p = (ColdArgs<B>)(base + sizeof(void*));
  asm volatile("" : "+s"(p));  // (opaque: every use site issues its own scalar load)
  return p;
}

// ---- device-scope primitives of the robot-migration queue (solo_kernel_params.h: MigrationQueue) -----------------
// relaxed atomics at agent scope (they go to the L2 / memory side: visible to every CU of every XCD) ...
__device__ __forceinline__ int wave_atomic_add(int32_t* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int wave_atomic_load(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wave_atomic_store(int32_t* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ... for the counters AND for the data a robot travels as (its record, its termination counters): every access to
// them in a migrating launch is a relaxed agent-scope atomic (sc1: served at the coherence point, whichever XCD asks -
// the L2s of the eight XCDs are not coherent with each other for plain accesses).  The publication is then ordered by
// waits alone: the data stores, s_waitcnt vmcnt(0) (a workgroup-scope release: no cache operation), the ring slot; and
// on the other side the poll, then the data loads.  (An agent-scope release / acquire PAIR of fences instead writes
// back and invalidates a whole L2 per hand-over: measured, 4096 hand-overs cost 90 us - a fifth of a 20-step launch.)
__device__ __forceinline__ float wave_load_shared(const float* p) { return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ double wave_load_shared(const double* p) { return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ void wave_store_shared(float* p, float v) { __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wave_store_shared(double* p, double v) { __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// (the workgroup-scope fence is a COMPILER barrier only - on gfx950 it emits no instruction -, so the wait is written
// out: every store and atomic this wave has issued has been acknowledged by the coherence point before anything behind
// this line is issued.  tests/test_kernel_asm.py looks for it in front of every ring-slot publication.)
__device__ __forceinline__ void wave_release_device() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void wave_acquire_device() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
__device__ __forceinline__ void wave_backoff() { __builtin_amdgcn_s_sleep(8); }
// the engine's fault word lives in pinned host memory: a system-scope store
__device__ __forceinline__ void wave_fault_set(int32_t* p) { __hip_atomic_store(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// the XCD this wave runs on (hardware register XCC_ID)
__device__ __forceinline__ int wave_xcc_id() { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf); }

// slot of this wave among the waves resident on its SIMD (HW_ID[3:0])
__device__ __forceinline__ int wave_slot_id() { return (int)(__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 4) & 0xf); }
// issue priority of this wave in its SIMD
__device__ __forceinline__ void wave_set_priority_level(int p) {  // p wave-uniform, 0..3
  if (p >= 3) __builtin_amdgcn_s_setprio(3);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else if (p == 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}

template <typename T> struct Real;
// f32: hardware v_sqrt / v_rsq / v_rcp (<= 1 ulp each) instead of the IEEE-correct library
// sequences (~10-25 instructions apiece), and a Cody-Waite + minimax sincos (~25 instructions,
// ~1 ulp for |x| < 1e4 rad; joint angles stay within +-20 rad) instead of libm's ~120-instruction
// sincosf with its Payne-Hanek path.  The f64 instantiation keeps the precise library versions:
// it is the parity path.
template <> struct Real<float> {
  static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
  static __device__ __forceinline__ float rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
  static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
  static constexpr int kTabSize = 0;  // (f32 coefficients are instruction literals)
  static __device__ __forceinline__ void sincos(float x, float* s, float* c, const float* = nullptr) {
    const float k = __builtin_rintf(x * 0.63661977236758134f);          // nearest multiple of pi/2
    float r = __builtin_fmaf(k, -1.57079601287841796875f, x);           // pi/2 in three pieces
    r = __builtin_fmaf(k, -3.1391647326017846e-07f, r);
    r = __builtin_fmaf(k, -5.3903029534742385e-15f, r);
    const float r2 = r * r;
    float sp = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = __builtin_fmaf(r2, sp, -1.6666654611e-1f);
    const float sn = __builtin_fmaf(r2 * r, sp, r);                     // sin(r), |r| <= pi/4
    float cp = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = __builtin_fmaf(r2, cp, 4.166664568298827e-2f);
    const float cs = __builtin_fmaf(r2 * r2, cp, __builtin_fmaf(r2, -0.5f, 1.0f));  // cos(r)
    const int q = (int)k;
    const float a = (q & 1) ? cs : sn, b = (q & 1) ? sn : cs;
    *s = (q & 2) ? -a : a;
    *c = ((q + 1) & 2) ? -b : b;
  }
  // sinc(x) = sin(x) / x and cos(x) from x^2, for the half rotation angle of one step (x = |w| dt / 2,
  // ~1e-2): even Taylor polynomials, exact to f32 round-off for x^2 < 1/16 (truncation < 5e-11 /
  // 3e-9 relative); beyond that (|w| > 500 rad/s at dt = 1e-3: never in a sane simulation) the
  // library path.  No sqrt, no range reduction: 8 fused multiply-adds instead of ~35 instructions.
  static __device__ __forceinline__ void sinc_cos(float x2, float* sinc, float* c, const float* = nullptr) {
    if (__builtin_expect(__builtin_amdgcn_readfirstlane(__float_as_int(x2)) > 0x3d800000, 0)) {  // x2 > 1/16 (wave-uniform: one robot)
      const float x = __builtin_amdgcn_sqrtf(x2);
      float sn, cs;
      sincos(x, &sn, &cs);
      *sinc = sn * __builtin_amdgcn_rcpf(x);
      *c = cs;
      return;
    }
    float sp = __builtin_fmaf(x2, -1.9841270e-4f, 8.3333333e-3f);
    sp = __builtin_fmaf(x2, sp, -1.6666667e-1f);
    *sinc = __builtin_fmaf(x2, sp, 1.0f);
    float cp = __builtin_fmaf(x2, 2.4801587e-5f, -1.3888889e-3f);
    cp = __builtin_fmaf(x2, cp, 4.1666667e-2f);
    cp = __builtin_fmaf(x2, cp, -0.5f);
    *c = __builtin_fmaf(x2, cp, 1.0f);
  }
  // atan2 by octant reduction + odd minimax polynomial on [0, 1] (max error 1.5e-7 rad), ~22
  // instructions instead of libm's ~60; asin(x) = atan2(x, sqrt(1 - x^2)).  These feed the Euler
  // angles of the observation / reward programs (obs.py:271, rewards.py:233).
  static __device__ __forceinline__ float atan2(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float a = (mx == 0.0f) ? 0.0f : mn * __builtin_amdgcn_rcpf(mx);
    const float s2 = a * a;
    float p = __builtin_fmaf(s2, -0.0040545654f, 0.021862952f);
    p = __builtin_fmaf(s2, p, -0.05591232f);
    p = __builtin_fmaf(s2, p, 0.09642197f);
    p = __builtin_fmaf(s2, p, -0.13908629f);
    p = __builtin_fmaf(s2, p, 0.19946566f);
    p = __builtin_fmaf(s2, p, -0.3332986f);
    p = __builtin_fmaf(s2, p, 0.99999934f);
    float r = a * p;
    r = (ay > ax) ? 1.57079637f - r : r;
    r = (x < 0.0f) ? 3.14159274f - r : r;
    return __builtin_copysignf(r, y);
  }
  static __device__ __forceinline__ float cos_of_asin(float x) { return __builtin_amdgcn_sqrtf(fmaxf(0.0f, __builtin_fmaf(-x, x, 1.0f))); }   // sqrt(1 - x^2)
  static __device__ __forceinline__ float asin(float x) { return atan2(x, cos_of_asin(x)); }
  // v_exp_f32 (2^x, 1 ulp) on a pre-scaled argument: the tolerance() rewards only need exp(-t^2/2), t^2/2 < 90
  static __device__ __forceinline__ float exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504f); }
  static __device__ __forceinline__ float abs(float x) { return fabsf(x); }
  static __device__ __forceinline__ float min(float a, float b) { return fminf(a, b); }
  static __device__ __forceinline__ float max(float a, float b) { return fmaxf(a, b); }
  static __device__ __forceinline__ float clamp(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
  // (a bit test, not isfinite(): the build may assume finite math for the arithmetic - see the Makefile -
  // and this check must keep seeing the NaNs / infinities of a diverged robot)
  static __device__ __forceinline__ bool finite(float x) { return (__float_as_uint(x) & 0x7f800000u) != 0x7f800000u; }
  static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
  static __device__ __forceinline__ float floor(float x) { return floorf(x); }
  static __device__ __forceinline__ float big() { return 3.0e38f; }
  static __device__ __forceinline__ float half_pi() { return 1.57079637f; }
  static __device__ __forceinline__ float half_ulp() { return 5.9604645e-8f; }  // 2^-24
};
// f64 (the reference's precision): hardware v_rsq_f64 / v_rcp_f64 seeds (~2^-26) refined by two fused
// Newton steps instead of the IEEE library sequences (division ~12, square root ~18, 1 / sqrt ~30
// instructions: ten of them per step), a Cody-Waite sincos with fdlibm's kernel polynomials (~35
// instructions, <= 2 ulp for |x| < 1e5 rad) instead of the library's sincos, whose Payne-Hanek path was
// the register peak of the whole kernel (52 of its 84 VGPR spills), and even Taylor polynomials in x^2 for the
// rotation update.  All within a few ulp of the correctly rounded value: three orders of magnitude below
// what the parity tests resolve (1e-13 per step against the oracle's different formulation).
template <> struct Real<double> {
  // 1 / sqrt(x): Goldschmidt iteration on g -> sqrt(x), h -> 1 / (2 sqrt(x)); returns 2 h
  static __device__ __forceinline__ void sqrt_pair(double x, double* g_out, double* h_out) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    *g_out = __builtin_fma(g, r, g);
    *h_out = __builtin_fma(h, r, h);
  }
  static __device__ __forceinline__ double sqrt(double x) {
    double g, h;
    sqrt_pair(x, &g, &h);
    return x == 0.0 ? 0.0 : g;  // (rsq(0) = inf: 0 x inf)
  }
  static __device__ __forceinline__ double rsqrt(double x) {
    double g, h;
    sqrt_pair(x, &g, &h);
    return h + h;  // (x <= 0 or non-finite: a non-finite result, which the diverged-robot guard sees)
  }
  static __device__ __forceinline__ double rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
  }
  // The polynomial coefficients come from a TABLE IN LDS (kMathTable below, staged once per launch): a f64
  // constant cannot be an instruction literal on gfx950 (VOP3 takes none), so each one is a register pair - written
  // as literals the compiler hoisted all ~30 of them out of the fused step loop and then SPILLED them (a scratch load
  // per coefficient per step); from LDS two coefficients arrive per ds_read_b128 broadcast.
  static constexpr int kTabSincos = 0, kTabSinc = 16, kTabSize = 32;
  static __device__ __forceinline__ void sincos(double x, double* s, double* c, const double* tab) {
    // r = x - k pi/2 with pi/2 in two pieces: the first fused step is exact for |k| < 2^20 (the difference
    // is a multiple of 2^-52 below 1), the second rounds once
    const double* t = tab + kTabSincos;
    const double k = __builtin_rint(x * t[0]);
    double r = __builtin_fma(k, t[1], x);
    r = __builtin_fma(k, t[2], r);
    const double z = r * r;
    // fdlibm __kernel_sin / __kernel_cos on [-pi/4, pi/4]
    double sp = __builtin_fma(z, t[4], t[5]);
    sp = __builtin_fma(z, sp, t[6]);
    sp = __builtin_fma(z, sp, t[7]);
    sp = __builtin_fma(z, sp, t[8]);
    sp = __builtin_fma(z, sp, t[9]);
    const double sn = __builtin_fma(z * r, sp, r);
    double cp = __builtin_fma(z, t[10], t[11]);
    cp = __builtin_fma(z, cp, t[12]);
    cp = __builtin_fma(z, cp, t[13]);
    cp = __builtin_fma(z, cp, t[14]);
    cp = __builtin_fma(z, cp, t[15]);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double cs = w + (((1.0 - w) - hz) + z * z * cp);  // (fdlibm's compensated 1 - z/2 + z^2 C(z))
    const int q = (int)k;
    const double a = (q & 1) ? cs : sn, b = (q & 1) ? sn : cs;
    *s = (q & 2) ? -a : a;
    *c = ((q + 1) & 2) ? -b : b;
  }
  // sinc(x) = sin(x) / x and cos(x) from x^2 (x = |w| dt / 2, ~1e-2): even Taylor polynomials, truncation
  // < 3e-21 / 4e-20 for x^2 < 1/16; beyond that (|w| > 500 rad/s at dt = 1e-3) the general path
  static __device__ __forceinline__ void sinc_cos(double x2, double* sinc, double* c, const double* tab) {
    if (__builtin_expect(__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(x2) >> 32)) > 0x3fb00000, 0)) {  // x2 > 1/16 (wave-uniform: one robot)
      const double x = sqrt(x2);
      double sn;
      sincos(x, &sn, c, tab);
      *sinc = sn * rcp(x);
      return;
    }
    const double* t = tab + kTabSinc;
    double sp = __builtin_fma(x2, t[0], t[1]);   // -1/15!, 1/13!
    sp = __builtin_fma(x2, sp, t[2]);            // -1/11!
    sp = __builtin_fma(x2, sp, t[3]);            // 1/9!
    sp = __builtin_fma(x2, sp, t[4]);            // -1/7!
    sp = __builtin_fma(x2, sp, t[5]);            // 1/5!
    sp = __builtin_fma(x2, sp, t[6]);            // -1/3!
    *sinc = __builtin_fma(x2, sp, 1.0);
    double cp = __builtin_fma(x2, t[8], t[9]);   // -1/14!, 1/12!
    cp = __builtin_fma(x2, cp, t[10]);           // -1/10!
    cp = __builtin_fma(x2, cp, t[11]);           // 1/8!
    cp = __builtin_fma(x2, cp, t[12]);           // -1/6!
    cp = __builtin_fma(x2, cp, t[13]);           // 1/4!
    cp = __builtin_fma(x2, cp, -0.5);
    *c = __builtin_fma(x2, cp, 1.0);
  }
  // The library atan2 / asin / exp (the Euler angles and the gaussian tolerance of the OUTPUT EPILOGUE only: once per
  // pass of <= 32 steps) are real CALLS: inlined, their ~40 f64 constants - register pairs, VOP3 takes no f64 literal -
  // were hoisted in front of the step / task loops and spilled there (48 SGPR spills in every f64 kernel; 42 scratch
  // stores per lane at the top of a migrating kernel: 24 KB per wave, 1190 B of HBM writes per env-step of a 20-step
  // launch, measured).  Behind a call the constants live and die inside the callee.
  static __device__ __attribute__((noinline)) double atan2(double y, double x) { return ::atan2(y, x); }
  // sqrt(1 - x^2) as sqrt((1 - x)(1 + x)): no cancellation near |x| = 1 (the Euler pitch: asin(x) = atan2(x, sqrt(1 - x^2)), solo_outputs.h)
  static __device__ __forceinline__ double cos_of_asin(double x) { return sqrt((1.0 - x) * (1.0 + x)); }
  static __device__ __forceinline__ double asin(double x) { return atan2(x, cos_of_asin(x)); }
  static __device__ __attribute__((noinline)) double exp(double x) { return ::exp(x); }
  static __device__ __forceinline__ double abs(double x) { return ::fabs(x); }
  static __device__ __forceinline__ double min(double a, double b) { return ::fmin(a, b); }
  static __device__ __forceinline__ double max(double a, double b) { return ::fmax(a, b); }
  static __device__ __forceinline__ double clamp(double x, double lo, double hi) { return ::fmin(::fmax(x, lo), hi); }
  static __device__ __forceinline__ bool finite(double x) { return ((unsigned long long)__double_as_longlong(x) & 0x7ff0000000000000ull) != 0x7ff0000000000000ull; }
  static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
  static __device__ __forceinline__ double floor(double x) { return ::floor(x); }
  // An f64 constant is a register pair (VOP3 takes no 64-bit literal), and one that is used inside the step loop is
  // hoisted in front of it and then SPILLED rather than rematerialised (two scratch stores per launch and a reload per
  // step for 1e300 and for pi/2 each: tools/spill_sites.py).  Built from two opaque halves it is materialised where it is
  // used: two v_mov.
  static __device__ __forceinline__ double pinned_constant(unsigned hi, unsigned lo) {
    asm volatile("" : "+v"(hi), "+v"(lo));
    return __hiloint2double((int)hi, (int)lo);
  }
  static __device__ __forceinline__ double big() { return pinned_constant(0x7e37e43cu, 0x8800759cu); }        // 1.0e300
  static __device__ __forceinline__ double half_pi() { return pinned_constant(0x3ff921fbu, 0x54442d18u); }    // 1.5707963267948966
  static __device__ __forceinline__ double half_ulp() { return 1.1102230246251565e-16; }  // 2^-53
};

// coefficients of Real<double>::sincos / sinc_cos (see there), copied into LDS by the step kernel's prologue
__device__ const double kMathTable[Real<double>::kTabSize] = {
    // [0] 2/pi, [1..2] -pi/2 in two pieces, [3] unused, [4..9] fdlibm S6..S1, [10..15] C6..C1
    6.36619772367581382433e-01, -1.57079632679489655800e+00, -6.12323399573676603587e-17, 0.0,
    1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06, -1.98412698298579493134e-04,
    8.33333333332248946124e-03, -1.66666666666666324348e-01,
    -1.13596475577881948265e-11, 2.08757232129817482790e-09, -2.75573143513906633035e-07, 2.48015872894767294178e-05,
    -1.38888888888741095749e-03, 4.16666666666666019037e-02,
    // [16..22] sinc: -1/15!, 1/13!, -1/11!, 1/9!, -1/7!, 1/5!, -1/3!; [23] unused; [24..29] cos: -1/14!, 1/12!, -1/10!, 1/8!, -1/6!, 1/4!
    -7.6471637318198164759e-13, 1.6059043836821614599e-10, -2.5052108385441718775e-08, 2.7557319223985890653e-06,
    -1.9841269841269841270e-04, 8.3333333333333333333e-03, -1.6666666666666666667e-01, 0.0,
    -1.1470745597729724714e-11, 2.0876756987868098979e-09, -2.7557319223985890653e-07, 2.4801587301587301587e-05,
    -1.3888888888888888889e-03, 4.1666666666666666667e-02, 0.0, 0.0};

template <typename T> __device__ __forceinline__ T wave_math_table(int i) { return T(kMathTable[i]); }

__device__ __forceinline__ void stats_add(double* p, double v) { atomicAdd(p, v); }

}  // namespace solo
