// MICROBENCH (not product): what does one link of the Gauss-Seidel row-update chain cost?
// Variants of the row-update block of solo_pgs_gfx950.h run back to back by one wave (iters times),
// each variant leaving one ingredient out; s_memtime ticks per block, one wave per SIMD / four per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x32 __attribute__((ext_vector_type(32)));

#define FETCH "s_set_gpr_idx_on %[rs], gpr_idx(SRC0)\n v_mov_b32_e32 %[col], v64\n s_set_gpr_idx_off\n"
#define NOFETCH "v_mov_b32_e32 %[col], v64\n"
#define READLANE "v_readlane_b32 %[sd], %[dl], %[rs]\n"
#define BODY_A "s_and_b64 %[todo], %[pend], %[w]\n v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n s_ff1_i32_b64 %[rn], %[todo]\n" \
               "v_fma_f32 %[v], %[sd], %[col], %[v]\n v_cndmask_b32_e32 %[lam], %[lam], %[cand], vcc\n v_med3_f32 %[cand], %[v], %[lo], %[hi]\n" \
               "v_mul_f32_e64 %[thr], %[tol], |%[lam]|\n v_sub_f32_e32 %[dl], %[cand], %[lam]\n s_cmp_eq_u32 %[rn], %[rn]\n"
#define CMP "v_cmp_gt_f32_e64 %[pend], |%[dl]|, %[thr]\n"
#define NOCMP "v_max_f32_e32 %[thr], %[dl], %[thr]\n"
#define TAIL "s_cbranch_scc0 9f\n s_lshl_b64 %[t], -2, %[rs]\n s_and_b64 %[t], %[w], %[t]\n s_and_b64 %[todo], %[pend], %[w]\n s_cbranch_scc0 9f\n s_ff1_i32_b64 %[rs], %[todo]\n"
#define TAIL_CONST_RS "s_cbranch_scc0 9f\n s_lshl_b64 %[t], -2, %[rs]\n s_and_b64 %[t], %[w], %[t]\n s_and_b64 %[todo], %[pend], %[w]\n s_cbranch_scc0 9f\n s_ff1_i32_b64 %[rn], %[todo]\n"

template <int MODE> __global__ __launch_bounds__(64, 4) void k(const float* in, float* out, unsigned long long* tt, int iters) {
  f32x32 a0, a1;
  for (int i = 0; i < 32; ++i) { a0[i] = in[threadIdx.x + 64 * i] * 1e-3f; a1[i] = in[threadIdx.x + 64 * (i + 32)] * 1e-3f; }
  float v = in[threadIdx.x], lam = 0.f, cand = v, dl = v, lo = -1e30f, hi = 1e30f, tol = 1e-30f, thr, col;
  int lane = threadIdx.x;
  unsigned long long pend = ~0ull, w = ~0ull, t, todo;
  int rs = 5, rn, sd, cnt = iters;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define RUN(TEXT)                                                                                        \
  asm volatile("1:\n" TEXT "9:\n s_sub_u32 %[cnt], %[cnt], 1\n s_cmp_lg_u32 %[cnt], 0\n s_cbranch_scc1 1b\n"       \
               : [v] "+v"(v), [lam] "+v"(lam), [cand] "+v"(cand), [dl] "+v"(dl), [pend] "+s"(pend), [thr] "=&v"(thr), [col] "=&v"(col), \
                 [t] "=&s"(t), [todo] "=&s"(todo), [rs] "+s"(rs), [rn] "=&s"(rn), [sd] "=&s"(sd), [cnt] "+s"(cnt)                \
               : [lane] "v"(lane), [tol] "v"(tol), [lo] "v"(lo), [hi] "v"(hi), [w] "s"(w), "{v[64:95]}"(a0), "{v[96:127]}"(a1)   \
               : "vcc", "scc", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63")
  if (MODE == 0) RUN(FETCH READLANE BODY_A CMP TAIL);                 // the real block (row index walks 0..63 via ff1 of all-ones & w -> always 0 here)
  if (MODE == 1) RUN(NOFETCH READLANE BODY_A CMP TAIL);               // no register indexing
  if (MODE == 2) RUN(FETCH "s_mov_b32 %[sd], 0x3a000000\n" BODY_A CMP TAIL);  // no v_readlane
  if (MODE == 3) RUN(FETCH READLANE BODY_A NOCMP TAIL);               // compare result not written to SGPRs
  if (MODE == 4) RUN(NOFETCH "s_mov_b32 %[sd], 0x3a000000\n" BODY_A NOCMP TAIL);  // none of the three
  if (MODE == 5) RUN(FETCH READLANE BODY_A CMP TAIL_CONST_RS);        // the next row index does not depend on this update's scalar chain
  if (MODE == 6) RUN("s_set_gpr_idx_on %[rs], gpr_idx(SRC0)\n s_set_gpr_idx_off\n");  // just the mode switches
  if (MODE == 7) RUN(READLANE "s_nop 0\n s_nop 0\n v_fma_f32 %[v], %[sd], %[v], %[v]\n v_sub_f32_e32 %[dl], %[v], %[lam]\n s_nop 0\n");  // readlane -> fma -> (sub) -> readlane round trip
  if (MODE == 9) RUN("s_ff1_i32_b64 %[rn], %[todo]\n v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n v_readlane_b32 %[sd], %[dl], %[rs]\n s_lshl_b64 %[t], -2, %[rs]\n s_set_gpr_idx_on %[rs], gpr_idx(SRC0)\n v_fma_f32 %[v], v64, %[sd], %[v]\n s_set_gpr_idx_off\n v_cndmask_b32_e32 %[lam], %[lam], %[cand], vcc\n v_med3_f32 %[cand], %[v], %[lo], %[hi]\n v_mul_f32_e64 %[thr], %[tol], |%[lam]|\n v_sub_f32_e32 %[dl], %[cand], %[lam]\n s_and_b64 %[t], %[w], %[t]\n v_cmp_gt_f32_e64 %[pend], |%[dl]|, %[thr]\n s_and_b64 %[todo], %[pend], %[t]\n s_cbranch_scc0 9f\n");  // the 15-instruction row update
  if (MODE == 10) RUN("s_ff1_i32_b64 %[rn], %[todo]\n v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n v_readlane_b32 %[sd], %[dl], %[rs]\n s_lshl_b64 %[t], -2, %[rs]\n s_set_gpr_idx_on %[rs], gpr_idx(SRC0)\n v_fma_f32 %[v], v64, %[sd], %[v]\n s_set_gpr_idx_off\n v_cndmask_b32_e32 %[lam], %[lam], %[cand], vcc\n v_med3_f32 %[cand], %[v], %[lo], %[hi]\n v_pk_fma_f32 v[60:61], v[56:57], v[58:59], v[62:63]\n s_and_b64 %[t], %[w], %[t]\n v_cmp_gt_f32_e64 %[pend], |v60|, |v61|\n s_and_b64 %[todo], %[pend], %[t]\n s_cbranch_scc0 9f\n");  // ... with one packed FMA for thr and dl (14)
  if (MODE == 12) RUN("s_ff1_i32_b64 %[rn], %[todo]\n v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n v_readlane_b32 %[sd], %[dl], %[rs]\n s_lshl_b64 %[t], -2, %[rs]\n s_set_gpr_idx_on %[rs], gpr_idx(SRC0)\n v_fma_f32 %[v], v64, %[sd], %[v]\n s_set_gpr_idx_off\n v_cndmask_b32_e32 %[lam], %[lam], %[cand], vcc\n v_med3_f32 %[cand], %[v], %[lo], %[hi]\n v_mul_f32_e32 %[thr], %[tol], %[lam]\n v_sub_f32_e32 %[dl], %[cand], %[lam]\n s_and_b64 %[t], %[w], %[t]\n v_cmp_gt_f32_e64 %[pend], |%[dl]|, |%[thr]|\n s_and_b64 %[todo], %[pend], %[t]\n s_cbranch_scc0 9f\n");  // thr = tol * lam (4-byte encoding), |thr| in the compare
  if (MODE == 8) RUN(CMP "s_and_b64 %[todo], %[pend], %[w]\n s_ff1_i32_b64 %[rn], %[todo]\n v_add_u32_e32 %[dl], %[rn], %[dl]\n");  // v_cmp -> SALU -> VALU round trip
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + threadIdx.x] = v + lam + cand + dl + (float)pend + (float)rs;
  if (threadIdx.x == 0) tt[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int blocks, const float* in) {
  float* out; unsigned long long* t; (void)hipMalloc(&out, blocks * 64 * 4); (void)hipMalloc(&t, blocks * 8);
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, in, out, t, iters); (void)hipDeviceSynchronize(); }
  unsigned long long* h = new unsigned long long[blocks]; (void)hipMemcpy(h, t, blocks * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < blocks; ++i) mean += h[i]; mean /= blocks;
  printf("%-52s waves/SIMD %3.1f: %7.1f ticks per block\n", name, blocks / 1024.0, mean / iters);
  delete[] h; (void)hipFree(out); (void)hipFree(t);
}
int main() {
  float* in; (void)hipMalloc(&in, 64 * 64 * 4);
  float h[64 * 64]; for (int i = 0; i < 64 * 64; ++i) h[i] = 1.0f + (i % 7) * 0.1f;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int blocks : {1024, 4096}) {
    run<0>("full row-update block (21 instr + 3 loop)", blocks, in);
    run<1>("  without register indexing", blocks, in);
    run<2>("  without v_readlane", blocks, in);
    run<3>("  without v_cmp -> SGPR", blocks, in);
    run<4>("  without all three", blocks, in);
    run<5>("  full, next row index independent", blocks, in);
    run<6>("s_set_gpr_idx_on + off only (+3 loop)", blocks, in);
    run<7>("readlane, 2 nop, fma, sub, nop (+3 loop)", blocks, in);
    run<8>("v_cmp->s_and->s_ff1->v_add (+3 loop)", blocks, in);
    run<9>("15-instruction row update (+3 loop)", blocks, in);
    run<12>("  thr = tol * lam, |thr| in the compare", blocks, in);
    run<10>("14 with v_pk_fma for thr + dl (+3 loop)", blocks, in);
  }
  return 0;
}
