// MICROBENCH (not product): issue cost of v_fma_f32 vs v_pk_fma_f32 on gfx950 for one wave alone on
// a SIMD and for four waves sharing one, dependent chains vs independent streams.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ __launch_bounds__(64) void k(float* out, unsigned long long* t, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 1e-4f;
  float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
  f2 p0 = {a, a + 1}, p1 = {a + 2, a + 3}, p2 = {a + 4, a + 5}, p3 = {a + 6, a + 7};
  const f2 pb = {b, b}, pc = {c, c};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 8 dependent scalar fma (one chain)
#pragma unroll
      for (int j = 0; j < 8; ++j) x0 = __builtin_fmaf(x0, b, c);
    } else if (MODE == 1) {  // 8 independent-ish scalar fma (4 chains x 2)
#pragma unroll
      for (int j = 0; j < 2; ++j) { x0 = __builtin_fmaf(x0, b, c); x1 = __builtin_fmaf(x1, b, c); x2 = __builtin_fmaf(x2, b, c); x3 = __builtin_fmaf(x3, b, c); }
    } else if (MODE == 2) {  // 8 dependent pk fma
#pragma unroll
      for (int j = 0; j < 8; ++j) p0 = __builtin_elementwise_fma(p0, pb, pc);
    } else {  // 8 pk fma, 4 chains x 2
#pragma unroll
      for (int j = 0; j < 2; ++j) { p0 = __builtin_elementwise_fma(p0, pb, pc); p1 = __builtin_elementwise_fma(p1, pb, pc); p2 = __builtin_elementwise_fma(p2, pb, pc); p3 = __builtin_elementwise_fma(p3, pb, pc); }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int blocks) {
  float* out; unsigned long long* t; hipMalloc(&out, blocks * 64 * 4); hipMalloc(&t, blocks * 8);
  const int iters = 20000;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, t, iters); hipDeviceSynchronize();
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, t, iters); hipDeviceSynchronize();
  unsigned long long h[8]; hipMemcpy(h, t, 8 * 8, hipMemcpyDeviceToHost);
  printf("%-34s blocks %5d: %.2f cycles per instruction (wave 0)\n", name, blocks, (double)h[0] / (iters * 8.0));
  hipFree(out); hipFree(t);
}
int main() {
  for (int blocks : {256, 4096}) {  // 256 blocks: <= 1 wave per SIMD; 4096: 4 waves per SIMD
    run<0>("v_fma_f32 dependent chain", blocks);
    run<1>("v_fma_f32 4 independent chains", blocks);
    run<2>("v_pk_fma_f32 dependent chain", blocks);
    run<3>("v_pk_fma_f32 4 independent chains", blocks);
  }
  return 0;
}
