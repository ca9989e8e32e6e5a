// MICROBENCH (not product), second attempt at a stand-alone reproducer of the s_set_gpr_idx_on adjacency failure: the
// step kernel's row update VERBATIM, with the physical registers of a build that fails (accumulator v7) or passes
// (accumulator v8) - tools/gpu_hazard_probe.py showed that on the step kernel the failure follows the PARITY of the
// accumulator register of the indexed v_fma (odd: wrong, even: right), whatever the other registers are.
//   hipcc -O3 --offload-arch=gfx950 -DACC=7 -DCAND=8 -o hz2_v7 gpr_idx_hazard2.hip
// Every wave: v[64:127] = 0..63 (the "columns"), v12 = lane number (the "impulse change"), and ITERS times
//   rs = next row (LCG) ; the row update of solo_pgs_gfx950.h in round 2's order, registers as in the kernel ;
// acc (vACC) must end as (ITERS / 64) * 85344, v3 ("lam": goes through the v_cndmask behind s_set_gpr_idx_off) as 7.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef ACC
#define ACC 7
#endif
#ifndef CAND
#define CAND 8
#endif
#define S2(x) #x
#define S(x) S2(x)
#define VACC "v" S(ACC)
#define VCAND "v" S(CAND)
#ifdef SHADOW_NOP
#define SHADOW "s_nop 0\n\t"
#else
#define SHADOW
#endif
typedef float f32x32 __attribute__((ext_vector_type(32)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

__global__ __launch_bounds__(64, 4) void hazard2(float* __restrict__ out, int iters) {
  f32x32 a0, a1;
#pragma unroll
  for (int i = 0; i < 32; ++i) { a0[i] = (float)i; a1[i] = (float)(32 + i); }
  const int lane = (int)threadIdx.x;
  const float lanef = (float)lane;
  float acc, keep;
  int rs0 = __builtin_amdgcn_readfirstlane((int)blockIdx.x & 63);
  int it0 = __builtin_amdgcn_readfirstlane(iters - 1);
  asm volatile(
      "v_mov_b32 v40, %[lane]\n\t"
      "v_mov_b32 v12, %[lanef]\n\t"
      "v_mov_b32 " VACC ", 0\n\t"
      "v_mov_b32 v3, 0x40e00000\n\t"          // lam = 7.0
      "v_mov_b32 " VCAND ", 0x40e00000\n\t"   // cand = 7.0
      "v_mov_b32 v5, 0x3a83126f\n\t"          // lo
      "v_mov_b32 v4, 0x7149f2ca\n\t"          // hi
      "v_mov_b32 v9, 0x3a83126f\n\t"          // tol
      "s_mov_b32 s40, %[rs0]\n\t"
      "s_mov_b32 s41, %[it0]\n\t"
      "s_mov_b64 s[54:55], 0xc003c003\n\t"
      "s_branch .Lh2_%=_loop\n\t"
      ".p2align 6\n\t"
      ".fill 2, 4, 0xbf800000\n"
      ".Lh2_%=_loop:\n\t"
      "s_mul_i32 s40, s40, 5\n\t"
      "s_add_u32 s40, s40, 1\n\t"
      "s_and_b32 s40, s40, 63\n\t"
      // ---- the row update, round 2's order, the kernel's registers
      "v_cmp_eq_u32_e32 vcc, s40, v40\n\t"
      "v_readlane_b32 s66, v12, s40\n\t"
      "s_lshl_b64 s[60:61], -2, s40\n\t"
      "s_set_gpr_idx_on s40, gpr_idx(SRC0)\n\t"
      SHADOW
      "v_fma_f32 " VACC ", v64, s66, " VACC "\n\t"
      "s_set_gpr_idx_off\n\t"
      "v_cndmask_b32_e32 v3, v3, " VCAND ", vcc\n\t"
      "v_med3_f32 " VCAND ", " VACC ", v5, v4\n\t"
      "v_mul_f32_e64 v13, v9, |v3|\n\t"
      "v_sub_f32_e32 v14, " VCAND ", v3\n\t"
      "s_and_b64 s[62:63], s[54:55], s[60:61]\n\t"
      "v_cmp_gt_f32_e64 s[10:11], |v14|, v13\n\t"
      "s_and_b64 s[62:63], s[10:11], s[62:63]\n\t"
      "v_mov_b32 " VCAND ", 0x40e00000\n\t"   // (cand back to 7.0: `lam` must stay 7 whichever lane is selected)
      "s_sub_u32 s41, s41, 1\n\t"
      "s_cbranch_scc0 .Lh2_%=_loop\n\t"
      "v_mov_b32 %[acc], " VACC "\n\t"
      "v_mov_b32 %[keep], v3\n\t"
      : [acc] "=&v"(acc), [keep] "=&v"(keep)
      : [lane] "v"(lane), [lanef] "v"(lanef), [rs0] "s"(rs0), [it0] "s"(it0), "{v[64:95]}"(a0), "{v[96:127]}"(a1)
      : "vcc", "scc", "v3", "v4", "v5", "v7", "v8", "v9", "v12", "v13", "v14", "v40", "s10", "s11", "s40", "s41", "s54", "s55", "s60", "s61", "s62", "s63", "s66");
  out[((size_t)blockIdx.x * 64 + lane) * 2] = acc;
  out[((size_t)blockIdx.x * 64 + lane) * 2 + 1] = keep;
}

int main() {
  const int iters = 8192, repeats = 4;
  const int grids[] = {1024, 2048, 4096, 8192};
  float* d_out = nullptr;
  CHECK(hipMalloc(&d_out, (size_t)8192 * 64 * 2 * sizeof(float)));
  std::vector<float> h((size_t)8192 * 64 * 2);
  const float want = (float)((long long)(iters / 64) * 85344ll);
  printf("accumulator v%d, cand v%d%s: expected acc %.0f, lam 7 in every lane\n", ACC, CAND,
#ifdef SHADOW_NOP
         ", s_nop behind s_set_gpr_idx_on",
#else
         "",
#endif
         want);
  for (int g : grids) {
    long long wrong_waves = 0, wrong_acc = 0, wrong_keep = 0;
    for (int r = 0; r < repeats; ++r) {
      CHECK(hipMemset(d_out, 0xff, (size_t)g * 64 * 2 * sizeof(float)));
      hipLaunchKernelGGL(hazard2, dim3(g), dim3(64), 0, 0, d_out, iters);
      CHECK(hipGetLastError());
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(h.data(), d_out, (size_t)g * 64 * 2 * sizeof(float), hipMemcpyDeviceToHost));
      for (int b = 0; b < g; ++b) {
        bool bad = false;
        for (int l = 0; l < 64; ++l) {
          if (!(h[((size_t)b * 64 + l) * 2] == want)) { ++wrong_acc; bad = true; }
          if (!(h[((size_t)b * 64 + l) * 2 + 1] == 7.0f)) { ++wrong_keep; bad = true; }
        }
        if (bad) ++wrong_waves;
      }
    }
    printf("  %5d waves x %d launches: %lld waves wrong (lanes: %lld wrong acc, %lld lam != 7)\n", g, repeats, wrong_waves, wrong_acc, wrong_keep);
  }
  CHECK(hipFree(d_out));
  return 0;
}
