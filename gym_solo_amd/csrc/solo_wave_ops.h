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

__device__ __forceinline__ float wave_readlane(float x, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane));
}
__device__ __forceinline__ double wave_readlane(double x, int lane) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float wave_shfl_xor(float x, int mask) { return __shfl_xor(x, mask, 64); }
__device__ __forceinline__ double wave_shfl_xor(double x, int mask) { return __shfl_xor(x, mask, 64); }
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __ballot(p); }

template <typename T> struct Real;
template <> struct Real<float> {
  static __device__ __forceinline__ float sqrt(float x) { return sqrtf(x); }
  static __device__ __forceinline__ float rsqrt(float x) { return 1.0f / sqrtf(x); }
  static __device__ __forceinline__ void sincos(float x, float* s, float* c) { sincosf(x, s, c); }
  static __device__ __forceinline__ float atan2(float y, float x) { return atan2f(y, x); }
  static __device__ __forceinline__ float asin(float x) { return asinf(x); }
  static __device__ __forceinline__ float exp(float x) { return expf(x); }
  static __device__ __forceinline__ float abs(float x) { return fabsf(x); }
  static __device__ __forceinline__ float min(float a, float b) { return fminf(a, b); }
  static __device__ __forceinline__ float max(float a, float b) { return fmaxf(a, b); }
  static __device__ __forceinline__ float clamp(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
  static __device__ __forceinline__ bool finite(float x) { return isfinite(x); }
  static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
  static __device__ __forceinline__ float big() { return 3.0e38f; }
};
template <> struct Real<double> {
  static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
  static __device__ __forceinline__ double rsqrt(double x) { return 1.0 / ::sqrt(x); }
  static __device__ __forceinline__ void sincos(double x, double* s, double* c) { ::sincos(x, s, c); }
  static __device__ __forceinline__ double atan2(double y, double x) { return ::atan2(y, x); }
  static __device__ __forceinline__ double asin(double x) { return ::asin(x); }
  static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
  static __device__ __forceinline__ double abs(double x) { return ::fabs(x); }
  static __device__ __forceinline__ double min(double a, double b) { return ::fmin(a, b); }
  static __device__ __forceinline__ double max(double a, double b) { return ::fmax(a, b); }
  static __device__ __forceinline__ double clamp(double x, double lo, double hi) { return ::fmin(::fmax(x, lo), hi); }
  static __device__ __forceinline__ bool finite(double x) { return isfinite(x); }
  static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
  static __device__ __forceinline__ double big() { return 1.0e300; }
};

__device__ __forceinline__ void stats_add(double* p, double v) { atomicAdd(p, v); }

}  // namespace solo
