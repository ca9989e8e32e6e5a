// MICROBENCH (not product): what the f64 Gauss-Seidel row update of solo_pgs_gfx950.h costs a wave that is ALONE on its
// SIMD (the slowest robot at the end of a launch), instruction by instruction, and what rearrangements of it would cost.
// Method of simd_rate.hip: a body of many straight-line repetitions, the same loop with an empty body subtracted, one
// wave per SIMD (1024 workgroups of 64), the median wave reported in cycles (s_memtime ticks) per repetition.
//
// build: hipcc -O3 --offload-arch=gfx950 -o pgs_row64 pgs_row64.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

enum Mode {
  kEmpty, kFma64Indep, kFma64Dep, kMax64Dep, kAdd64Dep, kMul64Indep, kCmp64, kReadlane2Fma,
  kRow19, kRowSaluMask, kRow18Even, kRowNoThr, kRowNoLam, kRowNoClamp, kRowNoIdx, kRowReordered, kRowReordered2, kRowNoCross, kRowConstLane, kRowConstDelta, kRowNops2, kRowNops4, kRowNops6, kRowPipelined, kModes
};
static const char* kName[kModes] = {
    "empty loop",
    "v_fma_f64 x4 independent",
    "v_fma_f64 one dependent chain",
    "v_max_f64 one dependent chain",
    "v_add_f64 one dependent chain",
    "v_mul_f64 x4 independent",
    "v_cmp_gt_f64 -> SGPR, independent",
    "readlane x2 -> v_fma_f64 (dependent through the scalar pair)",
    "the product's row update (19 instr; s_nop for the branch)",
    "  lane mask by s_lshl_b64 instead of v_cmp_eq_u32",
    "  even-lane layout (18 instr: no s_lshl_b32 for the register index)",
    "  without thr = tol |lam| (what the multiply costs)",
    "  without lam[row] = cand[row] (v_cmp_eq + 2 v_cndmask)",
    "  without the clamp (v_max + v_min)",
    "  without register indexing (plain column register)",
    "  reordered: scalar work under the VALU latencies",
    "  reordered 2: lam update behind the clamp",
    "  s_and todo reads a constant, not the compare's mask (VALU -> SALU crossing removed)",
    "  readlane lane select constant (SALU -> VALU crossing removed)",
    "  fma multiplies by a constant scalar pair (readlane -> fma removed)",
    "  + 2 s_nop 0 between the compare and the s_and",
    "  + 4 s_nop 0 between the compare and the s_and",
    "  + 6 s_nop 0 between the compare and the s_and",
    "  lam / thr update of the row moved BEHIND its compare (into the crossing's shadow)",
};
// instructions per repetition (for the per-instruction figure)
static const int kInstr[kModes] = {0, 4, 4, 4, 4, 4, 4, 3, 19, 19, 18, 18, 16, 17, 16, 19, 19, 19, 19, 19, 21, 23, 25, 19};
static const int kReps[kModes] = {0, 64, 64, 64, 64, 64, 64, 64, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16};

#define ROW_HEAD                                                                                                   \
  "s_ff1_i32_b64 %[rs], %[todo]\n"
#define ROW_TAIL                                                                                                   \
  "s_nop 0\n"                                                                                                      \
  "s_and_b64 %[todo], %[pend], %[w]\n s_or_b64 %[todo], %[todo], %[ph]\n"   /* (the s_or keeps the walk going: not counted as a row instruction, it replaces nothing) */

template <int MODE>
__device__ __forceinline__ void body(double& v, double& lam, double& cand, double& dl, double& x4, double& x5, double& x6, double& x7,
                                     double b, double c, int lane) {
  unsigned long long pend = 0, w, t, todo = 0x0000000009240924ull;
  const unsigned long long ph = 0x0000000009240924ull, ph_even = 0x0041041000410410ull;
  int rs, ri;
  double thr = c, c2 = 0;
  typedef double d16 __attribute__((ext_vector_type(16)));
  const d16 z0 = b * 1e-3, z1 = c;
  const double lo = -b, hi = b, tol = c;
  if (MODE == kFma64Indep) {
    asm volatile(".rept 64\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n .endr\n"
                 : "+v"(v), "+v"(lam), "+v"(cand), "+v"(dl) : "v"(b), "v"(c));
  } else if (MODE == kFma64Dep) {
    asm volatile(".rept 64\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n .endr\n"
                 : "+v"(v) : "v"(b), "v"(c));
  } else if (MODE == kMax64Dep) {
    asm volatile(".rept 64\n v_max_f64 %0, %0, %1\n v_min_f64 %0, %0, %2\n v_max_f64 %0, %0, %1\n v_min_f64 %0, %0, %2\n .endr\n"
                 : "+v"(v) : "v"(b), "v"(c));
  } else if (MODE == kAdd64Dep) {
    asm volatile(".rept 64\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, -%2\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, -%2\n .endr\n"
                 : "+v"(v) : "v"(b), "v"(c));
  } else if (MODE == kMul64Indep) {
    asm volatile(".rept 64\n v_mul_f64 %0, %4, |%5|\n v_mul_f64 %1, %4, |%5|\n v_mul_f64 %2, %4, |%5|\n v_mul_f64 %3, %4, |%5|\n .endr\n"
                 : "=&v"(v), "=&v"(lam), "=&v"(cand), "=&v"(dl) : "v"(b), "v"(c));
  } else if (MODE == kCmp64) {
    unsigned long long p1, p2, p3;
    asm volatile(".rept 64\n v_cmp_gt_f64_e64 %0, |%4|, %5\n v_cmp_gt_f64_e64 %1, |%4|, %5\n v_cmp_gt_f64_e64 %2, |%4|, %5\n v_cmp_gt_f64_e64 %3, |%4|, %5\n .endr\n"
                 : "=&s"(pend), "=&s"(p1), "=&s"(p2), "=&s"(p3) : "v"(b), "v"(c));
    x4 += (double)(pend + p1 + p2 + p3);
  } else if (MODE == kReadlane2Fma) {
    asm volatile(".rept 64\n v_readlane_b32 s94, v54, 3\n v_readlane_b32 s95, v55, 3\n v_fma_f64 v[54:55], %1, s[94:95], v[54:55]\n .endr\n"
                 : "+{v[54:55]}"(dl) : "v"(c) : "s94", "s95");
  } else if (MODE == kRow19) {
    asm volatile(".rept 16\n"
                 ROW_HEAD
                 "v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n"
                 "v_readlane_b32 s94, v54, %[rs]\n v_readlane_b32 s95, v55, %[rs]\n"
                 "s_lshl_b32 %[ri], %[rs], 1\n"
                 "s_set_gpr_idx_on %[ri], 1\n"
                 "s_lshl_b64 %[t], -2, %[rs]\n"
                 "v_fma_f64 %[v], v[64:65], s[94:95], %[v]\n"
                 "s_set_gpr_idx_off\n"
                 "s_and_b64 %[w], %[ph], %[t]\n"
                 "v_cndmask_b32_e32 v50, v50, v52, vcc\n v_cndmask_b32_e32 v51, v51, v53, vcc\n"
                 "v_max_f64 v[52:53], %[v], %[lo]\n v_min_f64 v[52:53], v[52:53], %[hi]\n"
                 "v_mul_f64 %[thr], %[tol], |v[50:51]|\n"
                 "v_add_f64 v[54:55], v[52:53], -v[50:51]\n"
                 "v_cmp_gt_f64_e64 %[pend], |v[54:55]|, %[thr]\n"
                 ROW_TAIL
                 ".endr\n"
                 : [v] "+v"(v), "+{v[50:51]}"(lam), "+{v[52:53]}"(cand), "+{v[54:55]}"(dl), [thr] "=&v"(thr), [pend] "+s"(pend), [w] "=&s"(w),
                   [t] "=&s"(t), [todo] "+s"(todo), [rs] "=&s"(rs), [ri] "=&s"(ri)
                 : [lane] "v"(lane), [tol] "v"(tol), [lo] "v"(lo), [hi] "v"(hi), [ph] "s"(ph), "{v[64:95]}"(z0), "{v[96:127]}"(z1)
                 : "vcc", "scc", "s94", "s95");
  } else if (MODE == kRowSaluMask) {
    asm volatile(".rept 16\n"
                 ROW_HEAD
                 "s_lshl_b64 vcc, 1, %[rs]\n"
                 "v_readlane_b32 s94, v54, %[rs]\n v_readlane_b32 s95, v55, %[rs]\n"
                 "s_lshl_b32 %[ri], %[rs], 1\n"
                 "s_set_gpr_idx_on %[ri], 1\n"
                 "s_lshl_b64 %[t], -2, %[rs]\n"
                 "v_fma_f64 %[v], v[64:65], s[94:95], %[v]\n"
                 "s_set_gpr_idx_off\n"
                 "s_and_b64 %[w], %[ph], %[t]\n"
                 "v_cndmask_b32_e32 v50, v50, v52, vcc\n v_cndmask_b32_e32 v51, v51, v53, vcc\n"
                 "v_max_f64 v[52:53], %[v], %[lo]\n v_min_f64 v[52:53], v[52:53], %[hi]\n"
                 "v_mul_f64 %[thr], %[tol], |v[50:51]|\n"
                 "v_add_f64 v[54:55], v[52:53], -v[50:51]\n"
                 "v_cmp_gt_f64_e64 %[pend], |v[54:55]|, %[thr]\n"
                 ROW_TAIL
                 ".endr\n"
                 : [v] "+v"(v), "+{v[50:51]}"(lam), "+{v[52:53]}"(cand), "+{v[54:55]}"(dl), [thr] "=&v"(thr), [pend] "+s"(pend), [w] "=&s"(w),
                   [t] "=&s"(t), [todo] "+s"(todo), [rs] "=&s"(rs), [ri] "=&s"(ri)
                 : [lane] "v"(lane), [tol] "v"(tol), [lo] "v"(lo), [hi] "v"(hi), [ph] "s"(ph), "{v[64:95]}"(z0), "{v[96:127]}"(z1)
                 : "vcc", "scc", "s94", "s95");
  } else if (MODE == kRow18Even) {
    todo = ph_even;
    asm volatile(".rept 16\n"
                 ROW_HEAD
                 "v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n"
                 "v_readlane_b32 s94, v54, %[rs]\n v_readlane_b32 s95, v55, %[rs]\n"
                 "s_set_gpr_idx_on %[rs], 1\n"
                 "s_lshl_b64 %[t], -2, %[rs]\n"
                 "v_fma_f64 %[v], v[64:65], s[94:95], %[v]\n"
                 "s_set_gpr_idx_off\n"
                 "s_and_b64 %[w], %[ph], %[t]\n"
                 "v_cndmask_b32_e32 v50, v50, v52, vcc\n v_cndmask_b32_e32 v51, v51, v53, vcc\n"
                 "v_max_f64 v[52:53], %[v], %[lo]\n v_min_f64 v[52:53], v[52:53], %[hi]\n"
                 "v_mul_f64 %[thr], %[tol], |v[50:51]|\n"
                 "v_add_f64 v[54:55], v[52:53], -v[50:51]\n"
                 "v_cmp_gt_f64_e64 %[pend], |v[54:55]|, %[thr]\n"
                 ROW_TAIL
                 ".endr\n"
                 : [v] "+v"(v), "+{v[50:51]}"(lam), "+{v[52:53]}"(cand), "+{v[54:55]}"(dl), [thr] "=&v"(thr), [pend] "+s"(pend), [w] "=&s"(w),
                   [t] "=&s"(t), [todo] "+s"(todo), [rs] "=&s"(rs)
                 : [lane] "v"(lane), [tol] "v"(tol), [lo] "v"(lo), [hi] "v"(hi), [ph] "s"(ph_even), "{v[64:95]}"(z0), "{v[96:127]}"(z1)
                 : "vcc", "scc", "s94", "s95");
  } else if (MODE == kRowNoThr) {
    thr = c;
    asm volatile(".rept 16\n"
                 ROW_HEAD
                 "v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n"
                 "v_readlane_b32 s94, v54, %[rs]\n v_readlane_b32 s95, v55, %[rs]\n"
                 "s_lshl_b32 %[ri], %[rs], 1\n"
                 "s_set_gpr_idx_on %[ri], 1\n"
                 "s_lshl_b64 %[t], -2, %[rs]\n"
                 "v_fma_f64 %[v], v[64:65], s[94:95], %[v]\n"
                 "s_set_gpr_idx_off\n"
                 "s_and_b64 %[w], %[ph], %[t]\n"
                 "v_cndmask_b32_e32 v50, v50, v52, vcc\n v_cndmask_b32_e32 v51, v51, v53, vcc\n"
                 "v_max_f64 v[52:53], %[v], %[lo]\n v_min_f64 v[52:53], v[52:53], %[hi]\n"
                 "v_add_f64 v[54:55], v[52:53], -v[50:51]\n"
                 "v_cmp_gt_f64_e64 %[pend], |v[54:55]|, %[thr]\n"
                 ROW_TAIL
                 ".endr\n"
                 : [v] "+v"(v), "+{v[50:51]}"(lam), "+{v[52:53]}"(cand), "+{v[54:55]}"(dl), [thr] "+v"(thr), [pend] "+s"(pend), [w] "=&s"(w),
                   [t] "=&s"(t), [todo] "+s"(todo), [rs] "=&s"(rs), [ri] "=&s"(ri)
                 : [lane] "v"(lane), [tol] "v"(tol), [lo] "v"(lo), [hi] "v"(hi), [ph] "s"(ph), "{v[64:95]}"(z0), "{v[96:127]}"(z1)
                 : "vcc", "scc", "s94", "s95");
  } else if (MODE == kRowNoLam) {
    asm volatile(".rept 16\n"
                 ROW_HEAD
                 "v_readlane_b32 s94, v54, %[rs]\n v_readlane_b32 s95, v55, %[rs]\n"
                 "s_lshl_b32 %[ri], %[rs], 1\n"
                 "s_set_gpr_idx_on %[ri], 1\n"
                 "s_lshl_b64 %[t], -2, %[rs]\n"
                 "v_fma_f64 %[v], v[64:65], s[94:95], %[v]\n"
                 "s_set_gpr_idx_off\n"
                 "s_and_b64 %[w], %[ph], %[t]\n"
                 "v_max_f64 v[52:53], %[v], %[lo]\n v_min_f64 v[52:53], v[52:53], %[hi]\n"
                 "v_mul_f64 %[thr], %[tol], |v[50:51]|\n"
                 "v_add_f64 v[54:55], v[52:53], -v[50:51]\n"
                 "v_cmp_gt_f64_e64 %[pend], |v[54:55]|, %[thr]\n"
                 ROW_TAIL
                 ".endr\n"
                 : [v] "+v"(v), "+{v[50:51]}"(lam), "+{v[52:53]}"(cand), "+{v[54:55]}"(dl), [thr] "=&v"(thr), [pend] "+s"(pend), [w] "=&s"(w),
                   [t] "=&s"(t), [todo] "+s"(todo), [rs] "=&s"(rs), [ri] "=&s"(ri)
                 : [lane] "v"(lane), [tol] "v"(tol), [lo] "v"(lo), [hi] "v"(hi), [ph] "s"(ph), "{v[64:95]}"(z0), "{v[96:127]}"(z1)
                 : "vcc", "scc", "s94", "s95");
  } else if (MODE == kRowNoClamp) {
    asm volatile(".rept 16\n"
                 ROW_HEAD
                 "v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n"
                 "v_readlane_b32 s94, v54, %[rs]\n v_readlane_b32 s95, v55, %[rs]\n"
                 "s_lshl_b32 %[ri], %[rs], 1\n"
                 "s_set_gpr_idx_on %[ri], 1\n"
                 "s_lshl_b64 %[t], -2, %[rs]\n"
                 "v_fma_f64 %[v], v[64:65], s[94:95], %[v]\n"
                 "s_set_gpr_idx_off\n"
                 "s_and_b64 %[w], %[ph], %[t]\n"
                 "v_cndmask_b32_e32 v50, v50, v52, vcc\n v_cndmask_b32_e32 v51, v51, v53, vcc\n"
                 "v_mul_f64 %[thr], %[tol], |v[50:51]|\n"
                 "v_add_f64 v[54:55], %[v], -v[50:51]\n"
                 "v_cmp_gt_f64_e64 %[pend], |v[54:55]|, %[thr]\n"
                 ROW_TAIL
                 ".endr\n"
                 : [v] "+v"(v), "+{v[50:51]}"(lam), "+{v[52:53]}"(cand), "+{v[54:55]}"(dl), [thr] "=&v"(thr), [pend] "+s"(pend), [w] "=&s"(w),
                   [t] "=&s"(t), [todo] "+s"(todo), [rs] "=&s"(rs), [ri] "=&s"(ri)
                 : [lane] "v"(lane), [tol] "v"(tol), [lo] "v"(lo), [hi] "v"(hi), [ph] "s"(ph), "{v[64:95]}"(z0), "{v[96:127]}"(z1)
                 : "vcc", "scc", "s94", "s95");
  } else if (MODE == kRowNoIdx) {
    asm volatile(".rept 16\n"
                 ROW_HEAD
                 "v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n"
                 "v_readlane_b32 s94, v54, %[rs]\n v_readlane_b32 s95, v55, %[rs]\n"
                 "s_lshl_b64 %[t], -2, %[rs]\n"
                 "v_fma_f64 %[v], v[64:65], s[94:95], %[v]\n"
                 "s_and_b64 %[w], %[ph], %[t]\n"
                 "v_cndmask_b32_e32 v50, v50, v52, vcc\n v_cndmask_b32_e32 v51, v51, v53, vcc\n"
                 "v_max_f64 v[52:53], %[v], %[lo]\n v_min_f64 v[52:53], v[52:53], %[hi]\n"
                 "v_mul_f64 %[thr], %[tol], |v[50:51]|\n"
                 "v_add_f64 v[54:55], v[52:53], -v[50:51]\n"
                 "v_cmp_gt_f64_e64 %[pend], |v[54:55]|, %[thr]\n"
                 ROW_TAIL
                 ".endr\n"
                 : [v] "+v"(v), "+{v[50:51]}"(lam), "+{v[52:53]}"(cand), "+{v[54:55]}"(dl), [thr] "=&v"(thr), [pend] "+s"(pend), [w] "=&s"(w),
                   [t] "=&s"(t), [todo] "+s"(todo), [rs] "=&s"(rs)
                 : [lane] "v"(lane), [tol] "v"(tol), [lo] "v"(lo), [hi] "v"(hi), [ph] "s"(ph), "{v[64:95]}"(z0), "{v[96:127]}"(z1)
                 : "vcc", "scc", "s94", "s95");
  } else if (MODE == kRowReordered) {
    // the scalar instructions that do not feed the chain (mask arithmetic) placed where the chain waits for a VALU result
    asm volatile(".rept 16\n"
                 ROW_HEAD
                 "v_readlane_b32 s94, v54, %[rs]\n v_readlane_b32 s95, v55, %[rs]\n"
                 "s_lshl_b32 %[ri], %[rs], 1\n"
                 "s_set_gpr_idx_on %[ri], 1\n"
                 "s_lshl_b64 %[t], -2, %[rs]\n"
                 "v_fma_f64 %[v], v[64:65], s[94:95], %[v]\n"
                 "s_set_gpr_idx_off\n"
                 "s_and_b64 %[w], %[ph], %[t]\n"
                 "v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n"
                 "v_max_f64 %[c2], %[v], %[lo]\n"
                 "v_cndmask_b32_e32 v50, v50, v52, vcc\n v_cndmask_b32_e32 v51, v51, v53, vcc\n"
                 "v_min_f64 v[52:53], %[c2], %[hi]\n"
                 "v_mul_f64 %[thr], %[tol], |v[50:51]|\n"
                 "v_add_f64 v[54:55], v[52:53], -v[50:51]\n"
                 "v_cmp_gt_f64_e64 %[pend], |v[54:55]|, %[thr]\n"
                 ROW_TAIL
                 ".endr\n"
                 : [v] "+v"(v), "+{v[50:51]}"(lam), "+{v[52:53]}"(cand), "+{v[54:55]}"(dl), [c2] "=&v"(c2), [thr] "=&v"(thr), [pend] "+s"(pend), [w] "=&s"(w),
                   [t] "=&s"(t), [todo] "+s"(todo), [rs] "=&s"(rs), [ri] "=&s"(ri)
                 : [lane] "v"(lane), [tol] "v"(tol), [lo] "v"(lo), [hi] "v"(hi), [ph] "s"(ph), "{v[64:95]}"(z0), "{v[96:127]}"(z1)
                 : "vcc", "scc", "s94", "s95");
  } else if (MODE == kRowReordered2) {
    // NOTE: writes cand (v_max) before lam takes the OLD cand: needs a second candidate register - here x4 holds the new one
    asm volatile(".rept 16\n"
                 ROW_HEAD
                 "v_readlane_b32 s94, v54, %[rs]\n v_readlane_b32 s95, v55, %[rs]\n"
                 "s_lshl_b32 %[ri], %[rs], 1\n"
                 "s_set_gpr_idx_on %[ri], 1\n"
                 "v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n"
                 "v_fma_f64 %[v], v[64:65], s[94:95], %[v]\n"
                 "s_set_gpr_idx_off\n"
                 "v_cndmask_b32_e32 v50, v50, v52, vcc\n v_cndmask_b32_e32 v51, v51, v53, vcc\n"
                 "s_lshl_b64 %[t], -2, %[rs]\n"
                 "v_max_f64 v[52:53], %[v], %[lo]\n"
                 "v_mul_f64 %[thr], %[tol], |v[50:51]|\n"
                 "v_min_f64 v[52:53], v[52:53], %[hi]\n"
                 "s_and_b64 %[w], %[ph], %[t]\n"
                 "v_add_f64 v[54:55], v[52:53], -v[50:51]\n"
                 "v_cmp_gt_f64_e64 %[pend], |v[54:55]|, %[thr]\n"
                 ROW_TAIL
                 ".endr\n"
                 : [v] "+v"(v), "+{v[50:51]}"(lam), "+{v[52:53]}"(cand), "+{v[54:55]}"(dl), [thr] "=&v"(thr), [pend] "+s"(pend), [w] "=&s"(w),
                   [t] "=&s"(t), [todo] "+s"(todo), [rs] "=&s"(rs), [ri] "=&s"(ri)
                 : [lane] "v"(lane), [tol] "v"(tol), [lo] "v"(lo), [hi] "v"(hi), [ph] "s"(ph), "{v[64:95]}"(z0), "{v[96:127]}"(z1)
                 : "vcc", "scc", "s94", "s95");
  } else if (MODE == kRowNoCross) {
    asm volatile(".rept 16\n"
                 ROW_HEAD
                 "v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n"
                 "v_readlane_b32 s94, v54, %[rs]\n v_readlane_b32 s95, v55, %[rs]\n"
                 "s_lshl_b32 %[ri], %[rs], 1\n"
                 "s_set_gpr_idx_on %[ri], 1\n"
                 "s_lshl_b64 %[t], -2, %[rs]\n"
                 "v_fma_f64 %[v], v[64:65], s[94:95], %[v]\n"
                 "s_set_gpr_idx_off\n"
                 "s_and_b64 %[w], %[ph], %[t]\n"
                 "v_cndmask_b32_e32 v50, v50, v52, vcc\n v_cndmask_b32_e32 v51, v51, v53, vcc\n"
                 "v_max_f64 v[52:53], %[v], %[lo]\n v_min_f64 v[52:53], v[52:53], %[hi]\n"
                 "v_mul_f64 %[thr], %[tol], |v[50:51]|\n"
                 "v_add_f64 v[54:55], v[52:53], -v[50:51]\n"
                 "v_cmp_gt_f64_e64 %[pend], |v[54:55]|, %[thr]\n"
                 
                 "s_nop 0\n"
                 "s_and_b64 %[todo], %[ph], %[w]\n s_or_b64 %[todo], %[todo], %[ph]\n"
                 ".endr\n"
                 : [v] "+v"(v), "+{v[50:51]}"(lam), "+{v[52:53]}"(cand), "+{v[54:55]}"(dl), [thr] "+v"(thr), [c2] "+v"(c2), [pend] "+s"(pend), [w] "=&s"(w),
                   [t] "=&s"(t), [todo] "+s"(todo), [rs] "=&s"(rs), [ri] "=&s"(ri)
                 : [lane] "v"(lane), [tol] "v"(tol), [lo] "v"(lo), [hi] "v"(hi), [ph] "s"(ph), [k3] "s"(3), [kd] "s"(c), "{v[64:95]}"(z0), "{v[96:127]}"(z1)
                 : "vcc", "scc", "s94", "s95");
  } else if (MODE == kRowConstLane) {
    asm volatile(".rept 16\n"
                 ROW_HEAD
                 "v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n"
                 "v_readlane_b32 s94, v54, %[k3]\n v_readlane_b32 s95, v55, %[k3]\n"
                 "s_lshl_b32 %[ri], %[rs], 1\n"
                 "s_set_gpr_idx_on %[ri], 1\n"
                 "s_lshl_b64 %[t], -2, %[rs]\n"
                 "v_fma_f64 %[v], v[64:65], s[94:95], %[v]\n"
                 "s_set_gpr_idx_off\n"
                 "s_and_b64 %[w], %[ph], %[t]\n"
                 "v_cndmask_b32_e32 v50, v50, v52, vcc\n v_cndmask_b32_e32 v51, v51, v53, vcc\n"
                 "v_max_f64 v[52:53], %[v], %[lo]\n v_min_f64 v[52:53], v[52:53], %[hi]\n"
                 "v_mul_f64 %[thr], %[tol], |v[50:51]|\n"
                 "v_add_f64 v[54:55], v[52:53], -v[50:51]\n"
                 "v_cmp_gt_f64_e64 %[pend], |v[54:55]|, %[thr]\n"
                 
                 "s_nop 0\n"
                 "s_and_b64 %[todo], %[pend], %[w]\n s_or_b64 %[todo], %[todo], %[ph]\n"
                 ".endr\n"
                 : [v] "+v"(v), "+{v[50:51]}"(lam), "+{v[52:53]}"(cand), "+{v[54:55]}"(dl), [thr] "+v"(thr), [c2] "+v"(c2), [pend] "+s"(pend), [w] "=&s"(w),
                   [t] "=&s"(t), [todo] "+s"(todo), [rs] "=&s"(rs), [ri] "=&s"(ri)
                 : [lane] "v"(lane), [tol] "v"(tol), [lo] "v"(lo), [hi] "v"(hi), [ph] "s"(ph), [k3] "s"(3), [kd] "s"(c), "{v[64:95]}"(z0), "{v[96:127]}"(z1)
                 : "vcc", "scc", "s94", "s95");
  } else if (MODE == kRowConstDelta) {
    asm volatile(".rept 16\n"
                 ROW_HEAD
                 "v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n"
                 "v_readlane_b32 s94, v54, %[rs]\n v_readlane_b32 s95, v55, %[rs]\n"
                 "s_lshl_b32 %[ri], %[rs], 1\n"
                 "s_set_gpr_idx_on %[ri], 1\n"
                 "s_lshl_b64 %[t], -2, %[rs]\n"
                 "v_fma_f64 %[v], v[64:65], %[kd], %[v]\n"
                 "s_set_gpr_idx_off\n"
                 "s_and_b64 %[w], %[ph], %[t]\n"
                 "v_cndmask_b32_e32 v50, v50, v52, vcc\n v_cndmask_b32_e32 v51, v51, v53, vcc\n"
                 "v_max_f64 v[52:53], %[v], %[lo]\n v_min_f64 v[52:53], v[52:53], %[hi]\n"
                 "v_mul_f64 %[thr], %[tol], |v[50:51]|\n"
                 "v_add_f64 v[54:55], v[52:53], -v[50:51]\n"
                 "v_cmp_gt_f64_e64 %[pend], |v[54:55]|, %[thr]\n"
                 
                 "s_nop 0\n"
                 "s_and_b64 %[todo], %[pend], %[w]\n s_or_b64 %[todo], %[todo], %[ph]\n"
                 ".endr\n"
                 : [v] "+v"(v), "+{v[50:51]}"(lam), "+{v[52:53]}"(cand), "+{v[54:55]}"(dl), [thr] "+v"(thr), [c2] "+v"(c2), [pend] "+s"(pend), [w] "=&s"(w),
                   [t] "=&s"(t), [todo] "+s"(todo), [rs] "=&s"(rs), [ri] "=&s"(ri)
                 : [lane] "v"(lane), [tol] "v"(tol), [lo] "v"(lo), [hi] "v"(hi), [ph] "s"(ph), [k3] "s"(3), [kd] "s"(c), "{v[64:95]}"(z0), "{v[96:127]}"(z1)
                 : "vcc", "scc", "s94", "s95");
  } else if (MODE == kRowNops2) {
    asm volatile(".rept 16\n"
                 ROW_HEAD
                 "v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n"
                 "v_readlane_b32 s94, v54, %[rs]\n v_readlane_b32 s95, v55, %[rs]\n"
                 "s_lshl_b32 %[ri], %[rs], 1\n"
                 "s_set_gpr_idx_on %[ri], 1\n"
                 "s_lshl_b64 %[t], -2, %[rs]\n"
                 "v_fma_f64 %[v], v[64:65], s[94:95], %[v]\n"
                 "s_set_gpr_idx_off\n"
                 "s_and_b64 %[w], %[ph], %[t]\n"
                 "v_cndmask_b32_e32 v50, v50, v52, vcc\n v_cndmask_b32_e32 v51, v51, v53, vcc\n"
                 "v_max_f64 v[52:53], %[v], %[lo]\n v_min_f64 v[52:53], v[52:53], %[hi]\n"
                 "v_mul_f64 %[thr], %[tol], |v[50:51]|\n"
                 "v_add_f64 v[54:55], v[52:53], -v[50:51]\n"
                 "v_cmp_gt_f64_e64 %[pend], |v[54:55]|, %[thr]\n"
                 "s_nop 0\n s_nop 0\n"
                 "s_nop 0\n"
                 "s_and_b64 %[todo], %[pend], %[w]\n s_or_b64 %[todo], %[todo], %[ph]\n"
                 ".endr\n"
                 : [v] "+v"(v), "+{v[50:51]}"(lam), "+{v[52:53]}"(cand), "+{v[54:55]}"(dl), [thr] "+v"(thr), [c2] "+v"(c2), [pend] "+s"(pend), [w] "=&s"(w),
                   [t] "=&s"(t), [todo] "+s"(todo), [rs] "=&s"(rs), [ri] "=&s"(ri)
                 : [lane] "v"(lane), [tol] "v"(tol), [lo] "v"(lo), [hi] "v"(hi), [ph] "s"(ph), [k3] "s"(3), [kd] "s"(c), "{v[64:95]}"(z0), "{v[96:127]}"(z1)
                 : "vcc", "scc", "s94", "s95");
  } else if (MODE == kRowNops4) {
    asm volatile(".rept 16\n"
                 ROW_HEAD
                 "v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n"
                 "v_readlane_b32 s94, v54, %[rs]\n v_readlane_b32 s95, v55, %[rs]\n"
                 "s_lshl_b32 %[ri], %[rs], 1\n"
                 "s_set_gpr_idx_on %[ri], 1\n"
                 "s_lshl_b64 %[t], -2, %[rs]\n"
                 "v_fma_f64 %[v], v[64:65], s[94:95], %[v]\n"
                 "s_set_gpr_idx_off\n"
                 "s_and_b64 %[w], %[ph], %[t]\n"
                 "v_cndmask_b32_e32 v50, v50, v52, vcc\n v_cndmask_b32_e32 v51, v51, v53, vcc\n"
                 "v_max_f64 v[52:53], %[v], %[lo]\n v_min_f64 v[52:53], v[52:53], %[hi]\n"
                 "v_mul_f64 %[thr], %[tol], |v[50:51]|\n"
                 "v_add_f64 v[54:55], v[52:53], -v[50:51]\n"
                 "v_cmp_gt_f64_e64 %[pend], |v[54:55]|, %[thr]\n"
                 "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
                 "s_nop 0\n"
                 "s_and_b64 %[todo], %[pend], %[w]\n s_or_b64 %[todo], %[todo], %[ph]\n"
                 ".endr\n"
                 : [v] "+v"(v), "+{v[50:51]}"(lam), "+{v[52:53]}"(cand), "+{v[54:55]}"(dl), [thr] "+v"(thr), [c2] "+v"(c2), [pend] "+s"(pend), [w] "=&s"(w),
                   [t] "=&s"(t), [todo] "+s"(todo), [rs] "=&s"(rs), [ri] "=&s"(ri)
                 : [lane] "v"(lane), [tol] "v"(tol), [lo] "v"(lo), [hi] "v"(hi), [ph] "s"(ph), [k3] "s"(3), [kd] "s"(c), "{v[64:95]}"(z0), "{v[96:127]}"(z1)
                 : "vcc", "scc", "s94", "s95");
  } else if (MODE == kRowNops6) {
    asm volatile(".rept 16\n"
                 ROW_HEAD
                 "v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n"
                 "v_readlane_b32 s94, v54, %[rs]\n v_readlane_b32 s95, v55, %[rs]\n"
                 "s_lshl_b32 %[ri], %[rs], 1\n"
                 "s_set_gpr_idx_on %[ri], 1\n"
                 "s_lshl_b64 %[t], -2, %[rs]\n"
                 "v_fma_f64 %[v], v[64:65], s[94:95], %[v]\n"
                 "s_set_gpr_idx_off\n"
                 "s_and_b64 %[w], %[ph], %[t]\n"
                 "v_cndmask_b32_e32 v50, v50, v52, vcc\n v_cndmask_b32_e32 v51, v51, v53, vcc\n"
                 "v_max_f64 v[52:53], %[v], %[lo]\n v_min_f64 v[52:53], v[52:53], %[hi]\n"
                 "v_mul_f64 %[thr], %[tol], |v[50:51]|\n"
                 "v_add_f64 v[54:55], v[52:53], -v[50:51]\n"
                 "v_cmp_gt_f64_e64 %[pend], |v[54:55]|, %[thr]\n"
                 "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
                 "s_nop 0\n"
                 "s_and_b64 %[todo], %[pend], %[w]\n s_or_b64 %[todo], %[todo], %[ph]\n"
                 ".endr\n"
                 : [v] "+v"(v), "+{v[50:51]}"(lam), "+{v[52:53]}"(cand), "+{v[54:55]}"(dl), [thr] "+v"(thr), [c2] "+v"(c2), [pend] "+s"(pend), [w] "=&s"(w),
                   [t] "=&s"(t), [todo] "+s"(todo), [rs] "=&s"(rs), [ri] "=&s"(ri)
                 : [lane] "v"(lane), [tol] "v"(tol), [lo] "v"(lo), [hi] "v"(hi), [ph] "s"(ph), [k3] "s"(3), [kd] "s"(c), "{v[64:95]}"(z0), "{v[96:127]}"(z1)
                 : "vcc", "scc", "s94", "s95");
  } else if (MODE == kRowPipelined) {
    asm volatile(".rept 16\n"
                 ROW_HEAD
                 ""
                 "v_readlane_b32 s94, v54, %[rs]\n v_readlane_b32 s95, v55, %[rs]\n"
                 "s_lshl_b32 %[ri], %[rs], 1\n"
                 "s_set_gpr_idx_on %[ri], 1\n"
                 "s_lshl_b64 %[t], -2, %[rs]\n"
                 "v_fma_f64 %[v], v[64:65], s[94:95], %[v]\n"
                 "s_set_gpr_idx_off\n"
                 "s_and_b64 %[w], %[ph], %[t]\n"
                 ""
                 "v_max_f64 %[c2], %[v], %[lo]\n v_min_f64 %[c2], %[c2], %[hi]\n"
                 ""
                 "v_add_f64 v[54:55], %[c2], -v[50:51]\n"
                 "v_cmp_gt_f64_e64 %[pend], |v[54:55]|, %[thr]\n"
                 "v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n"
                 "v_cndmask_b32_e32 v50, v50, v52, vcc\n v_cndmask_b32_e32 v51, v51, v53, vcc\n"
                 "v_mul_f64 %[thr], %[tol], |v[50:51]|\n"
                 "s_nop 0\n"
                 "s_and_b64 %[todo], %[pend], %[w]\n s_or_b64 %[todo], %[todo], %[ph]\n"
                 ".endr\n"
                 : [v] "+v"(v), "+{v[50:51]}"(lam), "+{v[52:53]}"(cand), "+{v[54:55]}"(dl), [thr] "+v"(thr), [c2] "+v"(c2), [pend] "+s"(pend), [w] "=&s"(w),
                   [t] "=&s"(t), [todo] "+s"(todo), [rs] "=&s"(rs), [ri] "=&s"(ri)
                 : [lane] "v"(lane), [tol] "v"(tol), [lo] "v"(lo), [hi] "v"(hi), [ph] "s"(ph), [k3] "s"(3), [kd] "s"(c), "{v[64:95]}"(z0), "{v[96:127]}"(z1)
                 : "vcc", "scc", "s94", "s95");
  }
  x5 += (double)todo;
}

template <int MODE>
__global__ __launch_bounds__(64, 4) void k(double* out, unsigned long long* ticks, int iters) {
  double v = threadIdx.x * 1e-3, lam = 1, cand = 2, dl = 3e-3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
  const double b = 1.0001, c = 1e-4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) body<MODE>(v, lam, cand, dl, x4, x5, x6, x7, b, c, (int)threadIdx.x);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + threadIdx.x] = v + lam + cand + dl + x4 + x5 + x6 + x7;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE> double run(int blocks, int iters) {
  double* out; unsigned long long* t;
  hipMalloc(&out, blocks * 64 * 8); hipMalloc(&t, blocks * 8);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, t, iters);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), t, blocks * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  hipFree(out); hipFree(t);
  return (double)h[blocks / 2];
}

template <int MODE> void report(double empty) {
  const int iters = 400;
  const double t = run<MODE>(1024, iters) - empty;
  printf("%-70s %7.1f cycles per repetition  (%5.2f per instruction, %d instr)\n", kName[MODE], t / (iters * (double)kReps[MODE]),
         t / (iters * (double)kReps[MODE] * kInstr[MODE]), kInstr[MODE]);
}

int main() {
  const double e = run<kEmpty>(1024, 400);
  printf("one wave per SIMD (1024 workgroups); empty loop %.1f ticks per iteration, subtracted\n", e / 400);
  report<kFma64Indep>(e); report<kFma64Dep>(e); report<kMax64Dep>(e); report<kAdd64Dep>(e); report<kMul64Indep>(e); report<kCmp64>(e);
  report<kReadlane2Fma>(e);
  report<kRow19>(e); report<kRowSaluMask>(e); report<kRow18Even>(e); report<kRowNoThr>(e); report<kRowNoLam>(e); report<kRowNoClamp>(e);
  report<kRowNoIdx>(e); report<kRowReordered>(e); report<kRowReordered2>(e);
  report<kRowNoCross>(e); report<kRowConstLane>(e); report<kRowConstDelta>(e); report<kRowNops2>(e); report<kRowNops4>(e); report<kRowNops6>(e); report<kRowPipelined>(e);
  return 0;
}
