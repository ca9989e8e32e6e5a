// MICROBENCH (not product): does the vector instruction directly behind s_set_gpr_idx_on / s_set_gpr_idx_off see the
// NEW index / mode?  An attempt at a stand-alone reproducer for the hazard met in round 3 inside the Gauss-Seidel loop
// (gym_solo_amd/csrc/solo_pgs_gfx950.h: wave-dependent garbage at two or more waves per SIMD in some builds of the
// step kernel, cured by ONE scalar instruction in each shadow).
//
// Every wave keeps v[64:127] = 0, 1, ... 63 and runs ITERS times the shape of the loop's row update:
//     rs = next row (a full-period LCG over 0..63: every row is visited ITERS / 64 times)
//     v_cmp_eq_u32 vcc, rs, lane
//     v_readlane_b32 sd, <lane number as float>, rs       ; sd = rs
//     s_set_gpr_idx_on rs, gpr_idx(SRC0)
//     [shadow 1]
//     v_fma_f32 acc, v64 (register-indexed: v[64 + rs]), sd, acc     ; acc += rs * rs
//     s_set_gpr_idx_off
//     [shadow 2]
//     v_cndmask_b32 keep, keep, keep2, vcc                 ; source 0 must be `keep` itself: stays 7.0 (keep2 = 7.0)
//     v_add_f32 chk, v64, chk                              ; source 0 must be v64 itself (= 0): chk stays 0
//     ... the rest of the product's row update on values of its own (v_med3, v_mul, v_sub, v_cmp into an SGPR pair,
//     s_and, taken branch), at the product loop's position within its 64-byte instruction line
// acc must end as (ITERS / 64) * sum r^2 = (ITERS / 64) * 85344 (exact in f32 for ITERS <= 8192), keep as 7 and chk as 0
// in every lane of every wave.  ORDERS: round 2's (the cursor shift before the mode switch, the indexed v_fma directly
// behind s_set_gpr_idx_on, the v_cndmask directly behind s_set_gpr_idx_off - the shape the compiler also emits for
// dynamically indexed arrays); the product's since round 3 (a scalar ALU instruction in each shadow); round 2's with
// s_nop 0 in the first, the second or both shadows.  CONTEXTS: 0 - every wave runs the loop; 1 - as in the step kernel
// the loop runs among other code: of every four consecutive waves one runs it at s_setprio 3, one at priority 0 and
// two run a mix of LDS round trips, transcendental and DPP instructions and global loads.  Grids put 1 ... 32 waves on
// every SIMD (4 resident at a time).  No address depends on a computed value: a wrong read cannot fault.
//
// build: hipcc -O3 --offload-arch=gfx950 -o gpr_idx_hazard gpr_idx_hazard.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x32 __attribute__((ext_vector_type(32)));

// the register of the accumulator - destination and third source of the indexed v_fma - can be pinned:
//   hipcc ... -DACC_REG=v7    (default: the compiler's choice; the kernels below happen to get an EVEN register)
#define HZ_STR2(x) #x
#define HZ_STR(x) HZ_STR2(x)
#ifdef ACC_REG
#define ACC_CONSTRAINT "+{" HZ_STR(ACC_REG) "}"
#define ACC_NAME HZ_STR(ACC_REG)
#else
#define ACC_CONSTRAINT "+v"
#define ACC_NAME "compiler's choice"
#endif

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

enum Order { kOld, kNew, kNopOn, kNopOff, kNopBoth, kOrders };
static const char* kOrderName[kOrders] = {
    "round 2's order: indexed v_fma directly behind s_set_gpr_idx_on, v_cndmask directly behind s_set_gpr_idx_off",
    "the product's order: a scalar ALU instruction in each shadow",
    "round 2's order + s_nop 0 behind s_set_gpr_idx_on",
    "round 2's order + s_nop 0 behind s_set_gpr_idx_off",
    "round 2's order + s_nop 0 in both shadows"};

#define HAZARD_LOOP(ON, OFF, TAIL)                                                                 \
  asm volatile(                                                                                    \
      "s_branch .Lhz_%=_loop\n\t"                                                                  \
      ".p2align 6\n\t"                                                                             \
      ".fill %[pad], 4, 0xbf800000\n"             /* (PAD = 12: the product loop's position in its 64-byte line) */ \
      ".Lhz_%=_loop:\n\t"                                                                          \
      "s_mul_i32 %[rs], %[rs], 5\n\t"                                                              \
      "s_add_u32 %[rs], %[rs], 1\n\t"                                                              \
      "s_and_b32 %[rs], %[rs], 63\n\t"                                                             \
      "v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n\t"                                                   \
      "v_readlane_b32 %[sd], %[lanef], %[rs]\n\t"                                                  \
      ON                                                                                           \
      "v_fma_f32 %[acc], v64, %[sd], %[acc]\n\t"                                                   \
      "s_set_gpr_idx_off\n\t"                                                                      \
      OFF                                                                                          \
      "v_cndmask_b32_e32 %[keep], %[keep], %[keep2], vcc\n\t"                                      \
      "v_add_f32 %[chk], v64, %[chk]\n\t"                                                          \
      "v_med3_f32 %[d1], %[acc], %[lo], %[hi]\n\t"                                                 \
      "v_mul_f32_e64 %[d2], %[lo], |%[keep]|\n\t"                                                  \
      "v_sub_f32_e32 %[d3], %[d1], %[keep]\n\t"                                                    \
      TAIL                                                                                         \
      "v_cmp_gt_f32_e64 %[pend], |%[d3]|, %[d2]\n\t"                                               \
      "s_and_b64 %[todo], %[pend], %[w]\n\t"                                                       \
      "s_sub_u32 %[it], %[it], 1\n\t"                                                              \
      "s_cbranch_scc0 .Lhz_%=_loop\n\t"           /* (borrow set when it goes below zero) */       \
      : [acc] ACC_CONSTRAINT(acc), [chk] "+v"(chk), [keep] "+v"(keep), [rs] "+s"(rs), [it] "+s"(it), [sd] "=&s"(sd), [t] "=&s"(t), \
        [w] "=&s"(w), [pend] "=&s"(pend), [todo] "=&s"(todo), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3) \
      : [lanef] "v"(lanef), [lane] "v"(lane), [keep2] "v"(keep2), [lo] "v"(lo), [hi] "v"(hi), [ph] "s"(ph), [pad] "i"(PAD), \
        "{v[64:95]}"(a0), "{v[96:127]}"(a1)                                                        \
      : "vcc", "scc")

#define SHIFT "s_lshl_b64 %[t], -2, %[rs]\n\t"
#define MASK "s_and_b64 %[w], %[ph], %[t]\n\t"
#define IDX_ON "s_set_gpr_idx_on %[rs], gpr_idx(SRC0)\n\t"

// what the neighbours of the loop execute in context 1: LDS round trips, transcendental and DPP instructions, loads
__device__ __forceinline__ float noise(const float* __restrict__ src, int lane, int iters) {
  __shared__ float s_buf[64];
  float x = (float)lane + 1.0f, y = 0.0f;
  for (int i = 0; i < iters; ++i) {
    s_buf[lane] = x;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    y += s_buf[(lane + i) & 63];
    x = __builtin_amdgcn_rcpf(x) + __builtin_amdgcn_sqrtf(y) + src[(i * 64 + lane) & 4095];
    x += __shfl_xor(x, 1);
  }
  return x + y;
}

template <int ORDER, int CONTEXT, int PAD>
__global__ __launch_bounds__(64, 4) void hazard_kernel(float* __restrict__ out, const float* __restrict__ src, int iters) {
  f32x32 a0, a1;
#pragma unroll
  for (int i = 0; i < 32; ++i) { a0[i] = (float)i; a1[i] = (float)(32 + i); }
  const int lane = (int)threadIdx.x;
  const float lanef = (float)lane;
  float* mine = out + ((size_t)blockIdx.x * 64 + lane) * 4;
  const int role = CONTEXT == 0 ? 1 : (int)(blockIdx.x & 3);
  if (role >= 2) {  // (context 1: a neighbour)
    mine[0] = noise(src, lane, iters / 8);
    mine[3] = -1.0f;
    return;
  }
  if (role == 0) __builtin_amdgcn_s_setprio(3);
  float acc = 0.0f, chk = 0.0f, keep = 7.0f;
  const float keep2 = 7.0f, lo = 1.0e-3f, hi = 1.0e30f;
  float d1, d2, d3;
  int rs = __builtin_amdgcn_readfirstlane((int)blockIdx.x & 63);  // (any start: the LCG has the full period)
  int it = __builtin_amdgcn_readfirstlane(iters - 1);
  int sd;
  unsigned long long t, w, pend, todo;
  const unsigned long long ph = 0xc003c003c003c003ull;
  if (ORDER == kOld) HAZARD_LOOP(SHIFT IDX_ON, "", MASK);
  else if (ORDER == kNew) HAZARD_LOOP(IDX_ON SHIFT, MASK, "");
  else if (ORDER == kNopOn) HAZARD_LOOP(SHIFT IDX_ON "s_nop 0\n\t", "", MASK);
  else if (ORDER == kNopOff) HAZARD_LOOP(SHIFT IDX_ON, "s_nop 0\n\t", MASK);
  else HAZARD_LOOP(SHIFT IDX_ON "s_nop 0\n\t", "s_nop 0\n\t", MASK);
  mine[0] = acc;
  mine[1] = chk;
  mine[2] = keep;
  mine[3] = (float)(todo & 1ull);  // (keeps the tail of the loop alive; 0 or 1)
}

struct Tally { long long waves, acc, chk, keep, measured; };

template <int ORDER, int CONTEXT, int PAD = 12>
static int run(int blocks, int iters, float* d_out, const float* d_src, std::vector<float>& h, Tally* tl) {
  CHECK(hipMemset(d_out, 0xff, (size_t)blocks * 64 * 4 * sizeof(float)));
  hipLaunchKernelGGL((hazard_kernel<ORDER, CONTEXT, PAD>), dim3(blocks), dim3(64), 0, 0, d_out, d_src, iters);
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h.data(), d_out, (size_t)blocks * 64 * 4 * sizeof(float), hipMemcpyDeviceToHost));
  const float want = (float)((long long)(iters / 64) * 85344ll);
  for (int b = 0; b < blocks; ++b) {
    if (CONTEXT == 1 && (b & 3) >= 2) continue;
    ++tl->measured;
    bool bad = false;
    for (int l = 0; l < 64; ++l) {
      const float* r = &h[((size_t)b * 64 + l) * 4];
      if (!(r[0] == want)) { ++tl->acc; bad = true; }
      if (!(r[1] == 0.0f)) { ++tl->chk; bad = true; }
      if (!(r[2] == 7.0f)) { ++tl->keep; bad = true; }
    }
    if (bad) ++tl->waves;
  }
  return 0;
}

template <int CONTEXT>
static int run_order(int order, int blocks, int iters, float* d_out, const float* d_src, std::vector<float>& h, Tally* tl) {
  switch (order) {
    case kOld: return run<kOld, CONTEXT>(blocks, iters, d_out, d_src, h, tl);
    case kNew: return run<kNew, CONTEXT>(blocks, iters, d_out, d_src, h, tl);
    case kNopOn: return run<kNopOn, CONTEXT>(blocks, iters, d_out, d_src, h, tl);
    case kNopOff: return run<kNopOff, CONTEXT>(blocks, iters, d_out, d_src, h, tl);
    default: return run<kNopBoth, CONTEXT>(blocks, iters, d_out, d_src, h, tl);
  }
}

template <int PAD>
static void sweep(int iters, int repeats, float* d_out, const float* d_src, std::vector<float>& h) {
  Tally tl = {0, 0, 0, 0, 0};
  for (int r = 0; r < repeats; ++r) (void)run<kOld, 1, PAD>(8192, iters, d_out, d_src, h, &tl);
  printf("   PAD %2d: %lld of %lld measured waves wrong (lanes: %lld wrong acc, %lld non-zero chk, %lld keep != 7)\n", PAD, tl.waves, tl.measured, tl.acc, tl.chk, tl.keep);
  if constexpr (PAD < 15) sweep<PAD + 1>(iters, repeats, d_out, d_src, h);
}

int main() {
  const int iters = 8192, repeats = 4;
  const int grids[] = {1024, 4096, 8192, 32768};
  const int max_blocks = 32768;
  float *d_out = nullptr, *d_src = nullptr;
  CHECK(hipMalloc(&d_out, (size_t)max_blocks * 64 * 4 * sizeof(float)));
  CHECK(hipMalloc(&d_src, 4096 * sizeof(float)));
  CHECK(hipMemset(d_src, 0, 4096 * sizeof(float)));
  std::vector<float> h((size_t)max_blocks * 64 * 4);
  printf("accumulator register of the indexed v_fma: %s\n", ACC_NAME);
  printf("every measured wave: %d row updates; expected acc = %lld, chk = 0, keep = 7 in every lane; 128 VGPRs per wave "
         "(4 waves per SIMD resident)\n", iters, (long long)(iters / 64) * 85344ll);
  for (int ctx = 0; ctx < 2; ++ctx) {
    printf("CONTEXT %d: %s\n", ctx, ctx == 0 ? "every wave runs the loop" : "of four consecutive waves one runs the loop at priority 3, one at priority 0, two run other code (LDS, transcendental, DPP, loads)");
    for (int o = 0; o < kOrders; ++o) {
      printf(" %s\n", kOrderName[o]);
      for (int g : grids) {
        Tally tl = {0, 0, 0, 0, 0};
        for (int r = 0; r < repeats; ++r) {
          const int rc = ctx == 0 ? run_order<0>(o, g, iters, d_out, d_src, h, &tl) : run_order<1>(o, g, iters, d_out, d_src, h, &tl);
          if (rc) return rc;
        }
        printf("   %6d waves x %d launches: %lld of %lld measured waves wrong (lanes: %lld wrong acc, %lld non-zero chk, %lld keep != 7)\n", g,
               repeats, tl.waves, tl.measured, tl.acc, tl.chk, tl.keep);
      }
    }
  }
  // the loop at every position within its 64-byte instruction line (round 2's order, context 1)
  printf("ALIGNMENT SWEEP: round 2's order, context 1, 8192 waves x %d launches, loop head PAD dwords past a 64-byte line\n", repeats);
  sweep<0>(iters, repeats, d_out, d_src, h);
  CHECK(hipFree(d_out));
  CHECK(hipFree(d_src));
  return 0;
}
