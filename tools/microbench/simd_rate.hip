// MICROBENCH (not product): VALU issue throughput of ONE SIMD vs the number of waves resident on it.
// Every wave runs the same stream of independent v_fma_f32 (inline asm: the compiler cannot pack or
// fold them) or a VALU / SALU mix; wave 0 reports s_memtime ticks per instruction.  The launch uses
// 64-thread workgroups: 1024 of them put one wave on every SIMD of the chip, 2048 two, ...
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ __launch_bounds__(64) void k(float* out, unsigned long long* t, int iters) {
  float x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
  const float b = 1.0001f, c = 1e-4f;
  int s0 = iters, s1 = 1, s2 = 2, s3 = 3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 8 independent v_fma
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(c));
    } else if (MODE == 1) {  // one dependent chain of 8
      asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                   "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                   : "+v"(x0) : "v"(b), "v"(c));
    } else if (MODE == 2) {  // 4 v_fma + 4 s_add interleaved (8 instructions)
      asm volatile("v_fma_f32 %0, %0, %8, %9\n s_add_u32 %4, %4, 1\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 %5, %5, 1\n"
                   "v_fma_f32 %2, %2, %8, %9\n s_add_u32 %6, %6, 1\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 %7, %7, 1\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(b), "v"(c) : "scc");
    } else if (MODE == 3) {  // 8 independent s_add
      asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                   "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                   : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
    } else if (MODE == 4) {  // 8 dependent v_add with DPP row_ror (the reductions' shape)
      asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                   "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                   : "+v"(x0));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + s0 + s1 + s2 + s3;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int blocks) {
  float* out; unsigned long long* t; hipMalloc(&out, blocks * 64 * 4); hipMalloc(&t, blocks * 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, t, iters); hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, t, iters);
  hipEventRecord(e1, 0); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long* h = new unsigned long long[blocks]; hipMemcpy(h, t, blocks * 8, hipMemcpyDeviceToHost);
  double mean = 0, mx = 0; for (int i = 0; i < blocks; ++i) { mean += h[i]; if (h[i] > mx) mx = h[i]; } mean /= blocks;
  printf("%-28s waves/SIMD %4.1f: %6.2f ticks per instruction per wave (mean; slowest wave %6.2f) ; launch %.3f ms -> %.2f ns per instr per wave, tick = %.3f ns\n",
         name, blocks / 1024.0, mean / (iters * 8.0), mx / (iters * 8.0), ms, ms * 1e6 / (iters * 8.0), ms * 1e6 / mx);
  delete[] h; hipFree(out); hipFree(t);
}
int main() {
  for (int blocks : {256, 1024, 2048, 4096, 8192}) {
    run<0>("v_fma 8 independent", blocks);
    run<1>("v_fma dependent chain", blocks);
    run<2>("v_fma / s_add interleaved", blocks);
    run<3>("s_add 4 chains", blocks);
    run<4>("v_add_dpp chain + s_nop 1", blocks);
  }
  return 0;
}
