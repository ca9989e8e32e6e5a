// MICROBENCH (not product): the 15-instruction Gauss-Seidel row-update loop of solo_pgs_gfx950.h at each of
// the 16 dword positions within a 64-byte instruction line: s_memtime ticks per iteration, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x32 __attribute__((ext_vector_type(32)));
template <int K> __global__ __launch_bounds__(64, 4) void k(const float* in, float* out, unsigned long long* tt, int iters) {
  f32x32 a0, a1;
  for (int i = 0; i < 32; ++i) { a0[i] = in[threadIdx.x + 64 * i] * 1e-3f; a1[i] = in[threadIdx.x + 64 * (i + 32)] * 1e-3f; }
  float v = in[threadIdx.x], lam = 0.f, cand = v, dl = v, lo = -1e30f, hi = 1e30f, tol = 1e-30f, thr;
  int lane = threadIdx.x;
  unsigned long long pend = ~0ull, w = ~0ull, t, todo = ~0ull;
  int rs = 5, rn, sd, cnt = iters;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_branch 2f\n .p2align 6\n .fill %[k], 4, 0xbf800000\n 2:\n 1:\n"
               "s_ff1_i32_b64 %[rn], %[todo]\n v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n v_readlane_b32 %[sd], %[dl], %[rs]\n s_lshl_b64 %[t], -2, %[rs]\n"
               "s_set_gpr_idx_on %[rs], gpr_idx(SRC0)\n v_fma_f32 %[v], v64, %[sd], %[v]\n s_set_gpr_idx_off\n v_cndmask_b32_e32 %[lam], %[lam], %[cand], vcc\n"
               "v_med3_f32 %[cand], %[v], %[lo], %[hi]\n v_mul_f32_e64 %[thr], %[tol], |%[lam]|\n v_sub_f32_e32 %[dl], %[cand], %[lam]\n s_and_b64 %[t], %[w], %[t]\n"
               "v_cmp_gt_f32_e64 %[pend], |%[dl]|, %[thr]\n s_sub_u32 %[cnt], %[cnt], 1\n s_cmp_lg_u32 %[cnt], 0\n s_cbranch_scc1 1b\n"
               : [v] "+v"(v), [lam] "+v"(lam), [cand] "+v"(cand), [dl] "+v"(dl), [pend] "+s"(pend), [thr] "=&v"(thr), [t] "=&s"(t), [todo] "+s"(todo),
                 [rs] "+s"(rs), [rn] "=&s"(rn), [sd] "=&s"(sd), [cnt] "+s"(cnt)
               : [lane] "v"(lane), [tol] "v"(tol), [lo] "v"(lo), [hi] "v"(hi), [w] "s"(w), [k] "n"(K), "{v[64:95]}"(a0), "{v[96:127]}"(a1)
               : "vcc", "scc");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + threadIdx.x] = v + lam + cand + dl + (float)pend + (float)rs;
  if (threadIdx.x == 0) tt[blockIdx.x] = t1 - t0;
}
template <int K> void run(const float* in) {
  const int blocks = 1024, iters = 20000;
  float* out; unsigned long long* t; (void)hipMalloc(&out, blocks * 64 * 4); (void)hipMalloc(&t, blocks * 8);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<K>, dim3(blocks), dim3(64), 0, 0, in, out, t, iters); (void)hipDeviceSynchronize(); }
  unsigned long long* h = new unsigned long long[blocks]; (void)hipMemcpy(h, t, blocks * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < blocks; ++i) mean += h[i]; mean /= blocks;
  printf("loop start %2d dwords past a 64-byte line: %6.1f ticks per iteration (16 instructions, 84 bytes)\n", K, mean / iters);
  delete[] h; (void)hipFree(out); (void)hipFree(t);
}
int main() {
  float* in; (void)hipMalloc(&in, 64 * 64 * 4);
  float h[64 * 64]; for (int i = 0; i < 64 * 64; ++i) h[i] = 1.0f + (i % 7) * 0.1f;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<0>(in); run<1>(in); run<2>(in); run<3>(in); run<4>(in); run<5>(in); run<6>(in); run<7>(in);
  run<8>(in); run<9>(in); run<10>(in); run<11>(in); run<12>(in); run<13>(in); run<14>(in); run<15>(in);
  return 0;
}
