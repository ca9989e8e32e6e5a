// MICROBENCH (not product): does a VALU instruction of a wave whose EXEC mask covers only some 16-lane rows take fewer issue
// cycles on gfx950?  (Does the SIMD skip the passes of rows that are switched off?)  Independent v_fma_f64 / v_fma_f32
// streams, 4 waves per SIMD (the VALU is the bottleneck), EXEC = all 64 lanes / the low 32 / the low 16.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, bool F64> __global__ __launch_bounds__(64, 4) void k(float* out, unsigned long long* tt, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.000001, c0 = 0.1, c1 = 0.2, c2 = 0.3, c3 = 0.4, c4 = 0.5, c5 = 0.6, c6 = 0.7, c7 = 0.8;
  float fa = threadIdx.x * 1e-3f, fb = 1.000001f, f0 = 0.1f, f1 = 0.2f, f2 = 0.3f, f3 = 0.4f, f4 = 0.5f, f5 = 0.6f, f6 = 0.7f, f7 = 0.8f;
  unsigned long long mask = MODE == 0 ? ~0ull : (MODE == 1 ? 0xffffffffull : 0xffffull);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (F64)
    asm volatile("s_mov_b64 exec, %[m]\n"
                 "1:\n"
                 "v_fma_f64 %[c0], %[a], %[b], %[c0]\n v_fma_f64 %[c1], %[a], %[b], %[c1]\n v_fma_f64 %[c2], %[a], %[b], %[c2]\n v_fma_f64 %[c3], %[a], %[b], %[c3]\n"
                 "v_fma_f64 %[c4], %[a], %[b], %[c4]\n v_fma_f64 %[c5], %[a], %[b], %[c5]\n v_fma_f64 %[c6], %[a], %[b], %[c6]\n v_fma_f64 %[c7], %[a], %[b], %[c7]\n"
                 "s_sub_u32 %[n], %[n], 1\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n"
                 "s_mov_b64 exec, -1\n"
                 : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [c4] "+v"(c4), [c5] "+v"(c5), [c6] "+v"(c6), [c7] "+v"(c7), [n] "+s"(iters)
                 : [a] "v"(a), [b] "v"(b), [m] "s"(mask) : "scc");
  else
    asm volatile("s_mov_b64 exec, %[m]\n"
                 "1:\n"
                 "v_fma_f32 %[c0], %[a], %[b], %[c0]\n v_fma_f32 %[c1], %[a], %[b], %[c1]\n v_fma_f32 %[c2], %[a], %[b], %[c2]\n v_fma_f32 %[c3], %[a], %[b], %[c3]\n"
                 "v_fma_f32 %[c4], %[a], %[b], %[c4]\n v_fma_f32 %[c5], %[a], %[b], %[c5]\n v_fma_f32 %[c6], %[a], %[b], %[c6]\n v_fma_f32 %[c7], %[a], %[b], %[c7]\n"
                 "s_sub_u32 %[n], %[n], 1\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n"
                 "s_mov_b64 exec, -1\n"
                 : [c0] "+v"(f0), [c1] "+v"(f1), [c2] "+v"(f2), [c3] "+v"(f3), [c4] "+v"(f4), [c5] "+v"(f5), [c6] "+v"(f6), [c7] "+v"(f7), [n] "+s"(iters)
                 : [a] "v"(fa), [b] "v"(fb), [m] "s"(mask) : "scc");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + threadIdx.x] = (float)(c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7) + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
  if (threadIdx.x == 0) tt[blockIdx.x] = t1 - t0;
}
template <int MODE, bool F64> void run(const char* name, int blocks) {
  float* out; unsigned long long* t; (void)hipMalloc(&out, blocks * 64 * 4); (void)hipMalloc(&t, blocks * 8);
  const int iters = 20000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) { (void)hipEventRecord(e0); hipLaunchKernelGGL((k<MODE, F64>), dim3(blocks), dim3(64), 0, 0, out, t, iters); (void)hipEventRecord(e1); (void)hipDeviceSynchronize(); (void)hipEventElapsedTime(&ms, e0, e1); }
  unsigned long long* h = new unsigned long long[blocks]; (void)hipMemcpy(h, t, blocks * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < blocks; ++i) mean += h[i]; mean /= blocks;
  printf("%-40s waves/SIMD %3.1f: %6.2f ticks per instruction per wave (kernel %.3f ms)\n", name, blocks / 1024.0, mean / iters / 8.0, ms);
  delete[] h; (void)hipFree(out); (void)hipFree(t);
}
int main() {
  for (int blocks : {1024, 4096}) {
    run<0, true>("v_fma_f64, EXEC = 64 lanes", blocks); run<1, true>("v_fma_f64, EXEC = low 32 lanes", blocks); run<2, true>("v_fma_f64, EXEC = low 16 lanes", blocks);
    run<0, false>("v_fma_f32, EXEC = 64 lanes", blocks); run<1, false>("v_fma_f32, EXEC = low 32 lanes", blocks); run<2, false>("v_fma_f32, EXEC = low 16 lanes", blocks);
  }
  return 0;
}
