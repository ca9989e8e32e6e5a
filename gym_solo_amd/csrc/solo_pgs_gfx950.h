// solo_pgs_gfx950.h — the projected Gauss-Seidel loop of solo_step_kernel.h (f32, register-resident
// Delassus columns) written directly in gfx950 assembly.  SAME rows, SAME order, SAME arithmetic as
// the C++ loop in physics_solve, which stays the definition: the f64 instantiation and the CPU wave
// emulator run it, and tests/test_gpu_pgs_asm.py compares this file against it BIT FOR BIT on the GPU
// (libsolo_hip_pgs_cpp.so: the same translation unit with -DSOLO_PGS_NO_ASM), sweep counts included.
//
// Why assembly.  The loop is a serial chain (a row's update decides which row is next), issued by ONE wave: what
// that costs was re-measured in round 3 (tools/microbench/simd_rate.hip, 256-instruction bodies:
// profiles/round3_microbench_simd_rate.log) - a lone wave issues INDEPENDENT instructions (v_fma_f32, s_add, s_nop)
// every 4.1 cycles, a dependent v_fma_f32 chain every 5.1, and this loop's row-update block every 6.2 cycles per
// instruction, whether its three neighbours on the SIMD have exited, sleep, or run at a lower priority; two
// independent row updates interleaved instruction by instruction would run at 5.2.  (Round 2's "7 cycles whatever
// the instruction" came from a microbenchmark that counted its 8-instruction loop's branch as measured work.)
// Gauss-Seidel offers no second independent row, so the time of this loop - half the time of the slowest robot of
// a launch, the one that decides how long the launch takes - follows its INSTRUCTION COUNT at ~6 cycles each, and
// that is what is minimised here:
//  * ONE 64-slot column bank.  The two 32-register tuples of ColumnBank<float> are pinned to
//    v[64:95] / v[96:127], so a column is `v64` indexed by the row number: three walks per sweep
//    instead of the compiler's six (one per phase and 32-register tuple);
//  * 15 instructions per updated row (compiler: 17): the column is not fetched into a register - the
//    FMA reads it register-indexed as its source 0 - and the rows beyond the cursor are the phase's
//    lanes above it (no window register to carry);
//  * 16 instructions per sweep besides them (compiler: ~70: flag registers for "a normal row moved",
//    mask halves moved about, 64-bit compares after instructions that had already set SCC, the
//    friction-limit refresh in 12 instructions instead of 8 with the DPP shifts folded into the
//    multiplies), with no "anything pending?" test in the loop.  The slowest robot of a closed-loop
//    step or a short launch runs all 50 sweeps with ~3 rows moving in each: one instruction per sweep
//    is 1 % of such a step, one per row 1.5 % (measured);
//  * few TAKEN branches on the slow robots' path (round 3): a taken branch costs a lone wave the refetch of its
//    instruction stream (~27 cycles) on top of its issue slot.  Replayed on the emulator, the robot-steps that run to
//    the sweep cap have, beyond their fifth sweep, work in (non-contact, normal, friction) rows = (no, yes, yes) in
//    91 % of the sweeps, and their walks update 1 / 2 / 3 / 4 / 6 rows in 30 / 37 / 9 / 14 / 8 % of the cases.  So
//    the sweep FALLS THROUGH into the normal rows when no motor / limit row moves (those rows are walked out of line)
//    and the row update is unrolled twice with the odd exit in the middle: ~2.7 taken branches per such sweep instead
//    of ~5.  Same-call A/B (profiles/round3_pgs_layout_ab.log): the driver's 20-step launch +1.9 % (kernel 0.383 ->
//    0.373 ms), one launch per step +3.1 %, f64 20-step kernel -2 %, 250-step rollouts +0.5 %;
//  * s_set_gpr_idx_on / s_set_gpr_idx_off are each FOLLOWED BY A SCALAR INSTRUCTION before the next vector instruction
//    (round 3).  With the indexed v_fma directly behind s_set_gpr_idx_on (and the v_cndmask directly behind
//    s_set_gpr_idx_off) the loop computed WAVE-DEPENDENT GARBAGE at two or more waves per SIMD in some builds - the
//    vector instruction saw the old index / mode: identical robots diverged, wild addresses faulted at 4096 robots,
//    a 64-robot batch (one wave per SIMD) was always right, and which build broke moved with the code around the
//    loop (rounds 2-3's product builds happened to be spared, bit for bit against the C++ loop; the first build
//    that re-entered the loop per sweep was not).  Four probes on the failing build, same call: s_nop after the
//    compare that writes `pend` - still faults; s_nop after the v_readlane - still faults; s_nop between
//    s_set_gpr_idx_on and the v_fma - clean; that plus the others - clean.  The fix costs nothing: the two scalar
//    instructions of the row update that do not depend on the vector pipe (s_lshl_b64 for the cursor, s_and_b64 for
//    the rows beyond it) sit in the two shadows.  tests/test_gpu_physics.py::test_identical_robots_stay_identical
//    guards it.  (The compiler's own s_set_gpr_idx_on / v_mov / s_set_gpr_idx_off sequences - dynamically indexed
//    local arrays - have the same shape: the kernel no longer contains any.)
//    Probed on the step kernel itself (tools/gpu_hazard_probe.py, -DSOLO_PGS_HAZARD_PROBE=n builds of this file: the
//    ROW variants below; profiles/round3_hazard_probe.log): only the adjacency s_set_gpr_idx_on -> indexed v_fma
//    matters (s_nop or the cursor shift behind s_set_gpr_idx_on cure it; s_nop in front of it or behind
//    s_set_gpr_idx_off do not); the compiler's shape - an indexed v_mov_b32 behind the switch - passes; and with the
//    accumulator of `v_fma_f32 vD, v[64 + idx], s, vD` pinned, vD = v6 / v8 pass and v7 / v9 / v11 FAIL, in every
//    kernel: round 2's product was right because the allocator had picked v8.  Stand-alone loops of the same shape,
//    down to the register numbers (tools/microbench/gpr_idx_hazard*.hip), never fail: the mechanism is not established.
//    Round 6 (one time-boxed session: profiles/round6_hazard_probe.log; probes 11-14 below, accumulator pinned to v7, the register
//    that fails): it is an ISSUE-SLOT hazard of the switch, not a property of scalar instructions - ONE instruction of any kind
//    between s_set_gpr_idx_on and the instruction that uses the index cures it, a v_nop exactly like an s_nop (probe 11: clean);
//    time in FRONT of the switch is irrelevant (s_nop 7 there: fails); the indexed operand as source 1 (gpr_idx(SRC1)) fails
//    alike, the VOP2 encoding v_fmac_f32 too (into memory faults); and it needs two waves on the SIMD: at 1024 robots - one wave
//    per SIMD but for the few SIMDs the dispatcher gives two - 0.3 ... 2.5 % of the robots go wrong, at 2048 (two per SIMD)
//    nearly all.  That is what a missing interlock between the switch's write of M0 / MODE and the operand fetch of the wave's
//    NEXT vector instruction would look like when the SIMD issues that instruction back to back (a wave alone issues every >= 4
//    cycles and never does) - narrowed, not proven: no documentation here names it.
//    The rule here is the conservative one - no vector instruction in the shadow of either mode switch (and, checked on the
//    generated code by tools/check_gpr_idx.py: the region between a switch on and its switch off is straight-line - no label, no
//    branch, no s_waitcnt - and holds exactly one vector instruction).
//  * (round 5) WHERE the row update waits.  tools/microbench/pgs_row64.hip + gen_row64_scan.py (one wave per SIMD, the f64
//    row in a straight line, one filler instruction inserted at every position in turn; profiles/round5_microbench_*):
//    v_fma / v_max / v_add_f64 issue every 4.0 cycles, dependent or not, yet the 19-instruction row took 129 cycles, not
//    76 - and removing any one of its cross-pipe dependencies changed nothing.  The wait sits at two places: a SCALAR
//    instruction issued behind a vector instruction that WRITES AN SGPR (v_readlane, v_cmp) stalls until that write has
//    landed, ~16 cycles - the s_lshl / s_set_gpr_idx_on behind the readlanes, and the s_and behind the compare; four
//    v_nop behind the readlanes and three behind the compare are free.  Vector instructions do not wait there.  So the
//    row's vector work that does not feed the chain - lam[row] = cand[row], thr = tol |lam| - now sits BEHIND THE
//    READLANES, and the lane mask of the updated row comes from a scalar shift in front of them (a vector compare would
//    be one more SGPR write): 129 -> 117 cycles per row for a wave alone, bit for bit the same arithmetic.  Same call
//    (profiles/round5_row_order_ab_*.log, round5_scalar_lanemask_ab_f64.log): f64 20-step launch +2.7 %, one launch
//    per step +4.2 %, launches of 250 +0.3 % (+1.5 % more with the scalar mask: one VALU instruction less per row for
//    the waves that share a SIMD); f32 +2.9 / +4.6 / +1.9 %.  What is left - ~16 cycles behind the compare, 4 behind the
//    readlanes - has no independent vector work to take: every arrangement of the three fillers there are measures the
//    same (the deferred / ping-pong orders of the scan).
//  * the manual wait states of gfx940-class hardware are respected by construction (>= 2 instructions
//    between a VALU write of an SGPR / VCC and a VALU read of it, >= 2 between a VALU write and a DPP
//    read, >= 1 before a v_readlane of a freshly written VGPR) - the assembler does not check them
//    inside inline asm.
#pragma once

#include "solo_wave_ops.h"

#ifndef SOLO_PGS_NO_ASM  // (-DSOLO_PGS_NO_ASM: the test build that runs the C++ definition of the loop instead)
#define SOLO_PGS_GFX950 1
#endif

namespace solo {

#ifdef SOLO_STAMPS
#define SOLO_PGS_COUNT_ROW "s_add_u32 %[nch], %[nch], 1\n\t"
#else
#define SOLO_PGS_COUNT_ROW
#endif

// the walk over the pending rows of one phase (lanes PH; entry: %[todo] = pend & PH, non-zero): 15
// instructions per updated row (the column is not fetched: the FMA reads it register-indexed).  Rows are
// visited in ascending order, so the rows still to visit after row r are PH & (bits above r) - no window
// register to initialise per phase.  Between a VALU write of an SGPR / VCC and the VALU read of it sit two
// other instructions (the manual wait states of gfx940-class hardware; the assembler does not check
// inline asm).
// vcc = the updated row's lane, by a SCALAR shift (round 5; a vector compare before: -DSOLO_PGS_VECTOR_LANEMASK, the A/B
// build): one VALU instruction less per row for the waves that share the SIMD, the same issue slot for a wave alone
#ifdef SOLO_PGS_VECTOR_LANEMASK
#define SOLO_PGS_LANE_MASK "v_cmp_eq_u32_e32 vcc, %[rs], %[lane]\n\t"
#define SOLO_PGS_LANE_OPERAND(lane) [lane] "v"(lane),
#else
#define SOLO_PGS_LANE_MASK "s_lshl_b64 vcc, 1, %[rs]\n\t"
#define SOLO_PGS_LANE_OPERAND(lane)
#endif
#if defined(SOLO_PGS_HAZARD_PROBE) && SOLO_PGS_HAZARD_PROBE > 0
// DIAGNOSTIC builds only (tools/gpu_hazard_probe.py; never the product): round 2's order of the row update - the indexed
// v_fma directly behind s_set_gpr_idx_on, the v_cndmask directly behind s_set_gpr_idx_off - with an s_nop in the first
// (probe 2), the second (probe 3) or neither shadow (probe 1), to see on the step kernel itself which adjacency matters
// probes: 1 = round 2's order; 2 = + s_nop behind s_set_gpr_idx_on; 3 = + s_nop behind s_set_gpr_idx_off; 4 = + s_nop IN FRONT
// of s_set_gpr_idx_on; 5 = the cursor shift behind s_set_gpr_idx_on (the product's first shadow), the second shadow
// empty; 6 = the compiler's own shape - s_set_gpr_idx_on / indexed v_mov_b32 / s_set_gpr_idx_off - and the v_fma on the
// moved value; 7 = as 6 with s_nop behind s_set_gpr_idx_on
// round 6 (the time-boxed session VERDICT r5 asked for; tools/gpu_hazard_probe.py, profiles/round6_hazard_probe.log) - all with the
// accumulator pinned to v7, the register that FAILS in round 2's order: 11 = a VALU no-op (v_nop) in the shadow instead of a scalar
// one; 12 = the indexed column as SOURCE 1 (gpr_idx(SRC1), the broadcast in source 0); 13 = the two-operand encoding (VOP2
// v_fmac_f32: no third source operand to collect - the broadcast first moved into a VGPR); 14 = s_nop 7 in FRONT of the switch
// (time since the v_readlane, none behind the switch)
#if SOLO_PGS_HAZARD_PROBE >= 11 && SOLO_PGS_HAZARD_PROBE <= 14
#define SOLO_PGS_V_CONSTRAINT "+{v7}"
#define SOLO_PGS_SD_CONSTRAINT "=&{s64}"
#elif SOLO_PGS_HAZARD_PROBE == 8      /* round 2's order with the accumulator / the broadcast pinned to the registers of the kernels that PASS */
#define SOLO_PGS_V_CONSTRAINT "+{v8}"
#define SOLO_PGS_SD_CONSTRAINT "=&{s66}"
#elif SOLO_PGS_HAZARD_PROBE == 9    /* ... and to the registers of the kernels that FAIL */
#define SOLO_PGS_V_CONSTRAINT "+{v7}"
#define SOLO_PGS_SD_CONSTRAINT "=&{s64}"
#elif SOLO_PGS_HAZARD_PROBE >= 10   /* -DSOLO_PGS_PROBE_V=... -DSOLO_PGS_PROBE_SD=...: any pair */
#define SOLO_PGS_STR2(x) #x
#define SOLO_PGS_STR(x) SOLO_PGS_STR2(x)
#define SOLO_PGS_V_CONSTRAINT "+{" SOLO_PGS_STR(SOLO_PGS_PROBE_V) "}"
#define SOLO_PGS_SD_CONSTRAINT "=&{" SOLO_PGS_STR(SOLO_PGS_PROBE_SD) "}"
#endif
#define SOLO_PGS_PROBE_SHIFT "s_lshl_b64 %[t], -2, %[rs]\n\t"
#define SOLO_PGS_PROBE_IDX_ON "s_set_gpr_idx_on %[rs], gpr_idx(SRC0)\n\t"
#define SOLO_PGS_PROBE_FMA "v_fma_f32 %[v], v64, %[sd], %[v]\n\ts_set_gpr_idx_off\n\t"
#define SOLO_PGS_PROBE_MOV_FMA "v_mov_b32_e32 %[thr], v64\n\ts_set_gpr_idx_off\n\tv_fma_f32 %[v], %[thr], %[sd], %[v]\n\t"
#if SOLO_PGS_HAZARD_PROBE == 11
#define SOLO_PGS_PROBE_CORE SOLO_PGS_PROBE_SHIFT SOLO_PGS_PROBE_IDX_ON "v_nop\n\t" SOLO_PGS_PROBE_FMA
#elif SOLO_PGS_HAZARD_PROBE == 12
#define SOLO_PGS_PROBE_CORE SOLO_PGS_PROBE_SHIFT "s_set_gpr_idx_on %[rs], gpr_idx(SRC1)\n\t" "v_fma_f32 %[v], %[sd], v64, %[v]\n\ts_set_gpr_idx_off\n\t"
#elif SOLO_PGS_HAZARD_PROBE == 13
#define SOLO_PGS_PROBE_CORE SOLO_PGS_PROBE_SHIFT "v_mov_b32_e32 %[thr], %[sd]\n\t" SOLO_PGS_PROBE_IDX_ON "v_fmac_f32_e32 %[v], v64, %[thr]\n\ts_set_gpr_idx_off\n\t"
#elif SOLO_PGS_HAZARD_PROBE == 14
#define SOLO_PGS_PROBE_CORE SOLO_PGS_PROBE_SHIFT "s_nop 7\n\t" SOLO_PGS_PROBE_IDX_ON SOLO_PGS_PROBE_FMA
#elif SOLO_PGS_HAZARD_PROBE == 2
#define SOLO_PGS_PROBE_CORE SOLO_PGS_PROBE_SHIFT SOLO_PGS_PROBE_IDX_ON "s_nop 0\n\t" SOLO_PGS_PROBE_FMA
#elif SOLO_PGS_HAZARD_PROBE == 3
#define SOLO_PGS_PROBE_CORE SOLO_PGS_PROBE_SHIFT SOLO_PGS_PROBE_IDX_ON SOLO_PGS_PROBE_FMA "s_nop 0\n\t"
#elif SOLO_PGS_HAZARD_PROBE == 4
#define SOLO_PGS_PROBE_CORE SOLO_PGS_PROBE_SHIFT "s_nop 0\n\t" SOLO_PGS_PROBE_IDX_ON SOLO_PGS_PROBE_FMA
#elif SOLO_PGS_HAZARD_PROBE == 5
#define SOLO_PGS_PROBE_CORE SOLO_PGS_PROBE_IDX_ON SOLO_PGS_PROBE_SHIFT SOLO_PGS_PROBE_FMA
#elif SOLO_PGS_HAZARD_PROBE == 6
#define SOLO_PGS_PROBE_CORE SOLO_PGS_PROBE_SHIFT SOLO_PGS_PROBE_IDX_ON SOLO_PGS_PROBE_MOV_FMA
#elif SOLO_PGS_HAZARD_PROBE == 7
#define SOLO_PGS_PROBE_CORE SOLO_PGS_PROBE_SHIFT SOLO_PGS_PROBE_IDX_ON "s_nop 0\n\t" SOLO_PGS_PROBE_MOV_FMA
#else
#define SOLO_PGS_PROBE_CORE SOLO_PGS_PROBE_SHIFT SOLO_PGS_PROBE_IDX_ON SOLO_PGS_PROBE_FMA
#endif
#define SOLO_PGS_ROW(PH)                                                                           \
  "s_ff1_i32_b64 %[rs], %[todo]\n\t"                                                               \
  SOLO_PGS_LANE_MASK                                                                               \
  "v_readlane_b32 %[sd], %[dl], %[rs]\n\t"                                                         \
  SOLO_PGS_PROBE_CORE                                                                              \
  "v_cndmask_b32_e32 %[lam], %[lam], %[cand], vcc\n\t"                                             \
  "v_med3_f32 %[cand], %[v], %[lo], %[hi]\n\t"                                                     \
  "v_mul_f32_e64 %[thr], %[tol], |%[lam]|\n\t"                                                     \
  "v_sub_f32_e32 %[dl], %[cand], %[lam]\n\t"                                                       \
  "s_and_b64 %[w], " PH ", %[t]\n\t"                                                               \
  "v_cmp_gt_f32_e64 %[pend], |%[dl]|, %[thr]\n\t"                                                  \
  SOLO_PGS_COUNT_ROW                                                                               \
  "s_and_b64 %[todo], %[pend], %[w]\n\t"
#else
#define SOLO_PGS_ROW(PH)                                                                           \
  "s_ff1_i32_b64 %[rs], %[todo]\n\t"          /* the row to update (wave-uniform) */               \
  SOLO_PGS_LANE_MASK                                                                               \
  "v_readlane_b32 %[sd], %[dl], %[rs]\n\t"    /* the change of its impulse */                      \
  "v_cndmask_b32_e32 %[lam], %[lam], %[cand], vcc\n\t"   /* lam[row] = cand[row]: VECTOR work behind the readlane - a scalar */ \
  "v_mul_f32_e64 %[thr], %[tol], |%[lam]|\n\t"           /* instruction issued there waits for the readlane's SGPR write (see above) */ \
  "s_set_gpr_idx_on %[rs], gpr_idx(SRC0)\n\t"                                                      \
  "s_lshl_b64 %[t], -2, %[rs]\n\t"            /* (a scalar instruction BETWEEN the mode switch and the indexed VALU instruction: see above) */ \
  "v_fma_f32 %[v], v64, %[sd], %[v]\n\t"      /* v += column * change; the column is v[64 + row]: source 0, register-indexed */ \
  "s_set_gpr_idx_off\n\t"                                                                          \
  "s_and_b64 %[w], " PH ", %[t]\n\t"          /* the phase's rows beyond the cursor (and the wait state after the mode switch) */ \
  "v_med3_f32 %[cand], %[v], %[lo], %[hi]\n\t"                                                     \
  "v_sub_f32_e32 %[dl], %[cand], %[lam]\n\t"                                                       \
  "v_cmp_gt_f32_e64 %[pend], |%[dl]|, %[thr]\n\t"                                                  \
  SOLO_PGS_COUNT_ROW                                                                               \
  "s_and_b64 %[todo], %[pend], %[w]\n\t"
#endif

// the walk over a phase's pending rows.  A TAKEN branch costs a lone wave ~27 cycles of instruction refetch on top of
// its issue slot (tools/microbench/loop_align.hip: the 15-instruction row loop takes 120 ... 132 cycles per iteration,
// the same instructions in a straight line 93), a branch that falls through ~4: the row update is therefore unrolled
// twice, with the exit after an odd number of rows in the middle (walks of the robots that decide the length of a
// launch - replayed on the emulator, tools/analyse_slow_steps.py - update 1 / 2 / 3 / 4 / 6 rows in 30 / 37 / 9 / 14 /
// 8 % of the cases: 0.8 taken branches per walk instead of 1.5)
#define SOLO_PGS_WALK(P, PH)                                                                       \
  ".Lpgs_%=_" P "_row:\n\t"                                                                        \
  SOLO_PGS_ROW(PH)                                                                                 \
  "s_cbranch_scc0 .Lpgs_%=_" P "_out\n\t"                                                          \
  SOLO_PGS_ROW(PH)                                                                                 \
  "s_cbranch_scc1 .Lpgs_%=_" P "_row\n"                                                            \
  ".Lpgs_%=_" P "_out:\n\t"

// friction limits = mu x the normal impulse their contact holds NOW (the normal row sits one lane below its
// first friction row, two below the second: DPP row shifts folded into the multiply; %[thr] is the
// threshold of the current impulses - the last row update left it), only after a walk over normal rows
#define SOLO_PGS_LIMITS                                                                            \
  "v_mul_f32_dpp %[x1], %[lam], %[mu] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                    \
  "v_mul_f32_dpp %[x2], %[lam], %[mu] row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                    \
  "v_cndmask_b32_e64 %[x1], %[x2], %[x1], %[tan1]\n\t"                                             \
  "v_cndmask_b32_e64 %[lo], %[lo], -%[x1], %[tang]\n\t"                                            \
  "v_cndmask_b32_e64 %[hi], %[hi], %[x1], %[tang]\n\t"                                             \
  "v_med3_f32 %[cand], %[v], %[lo], %[hi]\n\t"                                                     \
  "v_sub_f32_e32 %[dl], %[cand], %[lam]\n\t"                                                       \
  "v_cmp_gt_f32_e64 %[pend], |%[dl]|, %[thr]\n\t"

#ifndef SOLO_PGS_V_CONSTRAINT
#define SOLO_PGS_V_CONSTRAINT "+v"
#define SOLO_PGS_SD_CONSTRAINT "=&s"
#endif
// Runs the sweeps.  In: v (candidates at lam = 0), lam = 0, cand = clamp(v), dl = cand - lam, pend = rows
// above the tolerance, lo / hi (friction rows: refreshed here from their normal row's impulse), the
// resident columns.  Out: lam (the impulses), returns the number of sweeps.
// A sweep costs 8 scalar instructions besides its row updates and the limit refresh (8): one `and` + one
// branch per phase and the sweep counter with the branch back.  There is no "anything
// pending?" test in the loop: a sweep starts on the NO-WORK path (labels a1, a2), whose phase tests lead
// to `done` when all three fail - a sweep that finds work in a phase continues on the other path (b1, b2),
// which ends in the counter.  One instruction per sweep is one per cent of a closed-loop step: the
// slowest robot of a step runs all 50 sweeps with three rows moving in each.
__device__ __forceinline__ int pgs_solve_gfx950(const ColumnBank<float>& A, float& v, float& lam, float& cand, float& dl,
                                                unsigned long long& pend, float& lo, float& hi, float tol, int lane, float mu,
                                                unsigned long long tan1_lanes, unsigned long long tangent_lanes,
                                                unsigned long long phase0, unsigned long long phase1, unsigned long long phase2,
                                                int iters, int& n_changed) {
  float thr, x1, x2;
  unsigned long long w, t, todo;
  int rs, sd, it;
  asm volatile(
      // The loops below sit at a FIXED position relative to the 64-byte instruction lines (the padding is
      // jumped over): a loop whose head lies 0..4 dwords past a 32-byte boundary takes 120 cycles per
      // iteration, 5..7 dwords past it 128..132 (tools/microbench/loop_align.hip: instructions are fetched
      // in 32-byte blocks, and a taken branch into the tail of a block gets little from its first fetch).
      // From this entry the branch targets of the path a slow robot takes - the sweep head and the row loops of the
      // normal and the friction rows - lie 4, 0 and 1 dwords past a boundary, the out-of-line non-contact walk starts
      // on one (its padding is never executed) and the normal rows behind it 4 dwords past one.  (Round 2 measured
      // sixteen entry positions on the closed loop: 1.10 ... 1.15e8 env-steps/s - without the pinning, every edit of
      // the code in front of the loop moved all timings.)
      "s_branch .Lpgs_%=_entry\n\t"
      ".p2align 6\n\t"
      ".fill 2, 4, 0xbf800000\n"               // (s_nop 0)
      ".Lpgs_%=_entry:\n\t"
      "s_sub_u32 %[it], 0, %[iters]\n\t"       // counts up to zero: the carry of the increment is "cap reached"
      "s_cbranch_scc0 .Lpgs_%=_done\n"          // (no sweeps allowed)
      ".Lpgs_%=_sweep:\n\t"
      // The FALL-THROUGH path is the one the slow robots take: in the late sweeps of a creeping robot the non-contact
      // rows (joint motors, joint limits) stand still and normal + friction rows move (91 % of the sweeps beyond the
      // fifth of the robot-steps that run to the cap; all three phases: 8 %), so a sweep that finds no work there
      // falls through both tests into the normal rows, and a sweep whose non-contact rows move takes that phase out
      // of line (p0 ... below): one taken branch per sweep - the way back - instead of two.
      "s_and_b64 %[todo], %[pend], %[ph0]\n\t"
      "s_cbranch_scc1 .Lpgs_%=_p0\n\t"
      // ---- all normal rows, then the friction limits (a sweep that has found no work so far)
      "s_and_b64 %[todo], %[pend], %[ph1]\n\t"
      "s_cbranch_scc0 .Lpgs_%=_a2\n"
      SOLO_PGS_WALK("q1", "%[ph1]")
      SOLO_PGS_LIMITS
      // ---- all friction rows
      ".Lpgs_%=_b2:\n\t"
      "s_and_b64 %[todo], %[pend], %[ph2]\n\t"
      "s_cbranch_scc0 .Lpgs_%=_next\n"
      ".Lpgs_%=_p2:\n"
      SOLO_PGS_WALK("p2", "%[ph2]")
      ".Lpgs_%=_next:\n\t"
      "s_add_u32 %[it], %[it], 1\n\t"
      "s_cbranch_scc0 .Lpgs_%=_sweep\n\t"       // (no carry: below the sweep cap)
      "s_branch .Lpgs_%=_done\n"
      ".Lpgs_%=_a2:\n\t"                        // (no work in the first two phases)
      "s_and_b64 %[todo], %[pend], %[ph2]\n\t"
      "s_cbranch_scc1 .Lpgs_%=_p2\n\t"
      "s_branch .Lpgs_%=_done\n"                 // (nothing pending at the start of a sweep: converged)
      // ---- out of line: the non-contact rows (joint motors, joint limits), leg by leg, then the normal rows of
      //      a sweep that has found work
      ".p2align 5\n"                            // (never executed: behind an unconditional branch)
      ".Lpgs_%=_p0:\n"
      SOLO_PGS_WALK("p0", "%[ph0]")
      "s_and_b64 %[todo], %[pend], %[ph1]\n\t"
      "s_cbranch_scc0 .Lpgs_%=_b2\n"            // no normal row moves in this sweep: the friction limits stand
      SOLO_PGS_WALK("p1", "%[ph1]")
      SOLO_PGS_LIMITS
      "s_branch .Lpgs_%=_b2\n"
      ".Lpgs_%=_done:\n\t"
      : [v] SOLO_PGS_V_CONSTRAINT(v), [lam] "+v"(lam), [cand] "+v"(cand), [dl] "+v"(dl), [lo] "+v"(lo), [hi] "+v"(hi), [pend] "+s"(pend),
        [thr] "=&v"(thr), [x1] "=&v"(x1), [x2] "=&v"(x2),
        [w] "=&s"(w), [t] "=&s"(t), [todo] "=&s"(todo), [rs] "=&s"(rs), [sd] SOLO_PGS_SD_CONSTRAINT(sd), [it] "=&s"(it)
#ifdef SOLO_STAMPS
        , [nch] "+s"(n_changed)
#endif
      : SOLO_PGS_LANE_OPERAND(lane) [tol] "v"(tol), [mu] "v"(mu), [iters] "s"(iters), [ph0] "s"(phase0), [ph1] "s"(phase1), [ph2] "s"(phase2),
        [tan1] "s"(tan1_lanes), [tang] "s"(tangent_lanes), "{v[64:95]}"(A.a0), "{v[96:127]}"(A.a1)
      : "vcc", "scc");
  // (s_set_gpr_idx_on overwrites M0.  M0 is a RESERVED register for the AMDGPU backend - naming it in the clobber
  // list draws "reserved registers on the clobber list may not be preserved ... undefined behaviour" from clang -
  // and the backend never keeps a value live in it across statements: it (re)writes M0 immediately before each
  // of its own uses - LDS-direct, s_movrel, sendmsg.  `make asm`: no m0 reference in the generated code.)
  (void)n_changed;
  return it + iters;
}

#undef SOLO_PGS_LIMITS
#undef SOLO_PGS_WALK
#undef SOLO_PGS_ROW

// ---- the same loop in f64: the reference's precision, the parity path (round 3; slot space since round 4) ----
// The f64 solver runs in SLOT space (ColumnBank<double>, solo_step_kernel.h: lane = slot, the live rows in lane order
// on the lanes 0 .. L-1) and this loop only sees steps with L <= 32: 32 columns per lane in v[104:167]
// (ColumnBank<double>: two 16-wide tuples), column r = v[104 + 2 r : 105 + 2 r], read register-indexed as source 0 of
// the v_fma_f64 (index 2 r).  The phases are lane masks of the step (operands, as before).  The 64-bit loop variables
// whose HALVES are touched (v_cndmask_b32 / v_readlane_b32 / DPP moves work on dwords) sit in fixed registers - an
// inline-asm operand cannot be sliced - : lam v[90:91], cand v[92:93], dl v[94:95], lo v[96:97], hi v[98:99],
// x1 v[100:101], x2 v[102:103]; the impulse change of the updated row in s[94:95].  The whole kernel then fits 168
// VGPRs = three waves per SIMD (round 3: 64 columns in v[128:255], 256 VGPRs, two waves).
// 19 instructions per updated row (the compiler's loop over LDS-evaluated columns: ~40), no v_med3 in f64:
// v_max_f64 + v_min_f64, exactly Real<double>::clamp.  Same rows, same order, same arithmetic as the C++ loop
// (tests/test_gpu_pgs_asm.py compares the two bit for bit in f64 too).
// The fixed registers of the f64 loop, by build: 128 VGPRs (FOUR waves per SIMD: the product since round 5) - loop variables
// v[50:63], columns v[64:127]; -DSOLO_F64_WAVES=3 / 2 (168 / 256 VGPRs: the A/B builds) - v[90:103], v[104:167]
#if !defined(SOLO_F64_WAVES) || SOLO_F64_WAVES >= 4
#define SOLO_PGS64_LAM "v[50:51]"
#define SOLO_PGS64_LAM_LO "v50"
#define SOLO_PGS64_LAM_HI "v51"
#define SOLO_PGS64_CAND "v[52:53]"
#define SOLO_PGS64_CAND_LO "v52"
#define SOLO_PGS64_CAND_HI "v53"
#define SOLO_PGS64_DL "v[54:55]"
#define SOLO_PGS64_DL_LO "v54"
#define SOLO_PGS64_DL_HI "v55"
#define SOLO_PGS64_LO "v[56:57]"
#define SOLO_PGS64_LO_LO "v56"
#define SOLO_PGS64_LO_HI "v57"
#define SOLO_PGS64_HI "v[58:59]"
#define SOLO_PGS64_HI_LO "v58"
#define SOLO_PGS64_HI_HI "v59"
#define SOLO_PGS64_X1 "v[60:61]"
#define SOLO_PGS64_X1_LO "v60"
#define SOLO_PGS64_X1_HI "v61"
#define SOLO_PGS64_X2 "v[62:63]"
#define SOLO_PGS64_X2_LO "v62"
#define SOLO_PGS64_X2_HI "v63"
#define SOLO_PGS64_COL0 "v[64:65]"
#define SOLO_PGS64_BANK0 "v[64:95]"
#define SOLO_PGS64_BANK1 "v[96:127]"
#else
#define SOLO_PGS64_LAM "v[90:91]"
#define SOLO_PGS64_LAM_LO "v90"
#define SOLO_PGS64_LAM_HI "v91"
#define SOLO_PGS64_CAND "v[92:93]"
#define SOLO_PGS64_CAND_LO "v92"
#define SOLO_PGS64_CAND_HI "v93"
#define SOLO_PGS64_DL "v[94:95]"
#define SOLO_PGS64_DL_LO "v94"
#define SOLO_PGS64_DL_HI "v95"
#define SOLO_PGS64_LO "v[96:97]"
#define SOLO_PGS64_LO_LO "v96"
#define SOLO_PGS64_LO_HI "v97"
#define SOLO_PGS64_HI "v[98:99]"
#define SOLO_PGS64_HI_LO "v98"
#define SOLO_PGS64_HI_HI "v99"
#define SOLO_PGS64_X1 "v[100:101]"
#define SOLO_PGS64_X1_LO "v100"
#define SOLO_PGS64_X1_HI "v101"
#define SOLO_PGS64_X2 "v[102:103]"
#define SOLO_PGS64_X2_LO "v102"
#define SOLO_PGS64_X2_HI "v103"
#define SOLO_PGS64_COL0 "v[104:105]"
#define SOLO_PGS64_BANK0 "v[104:135]"
#define SOLO_PGS64_BANK1 "v[136:167]"
#endif
#define SOLO_PGS_ROW64(PH)                                                                         \
  "s_ff1_i32_b64 %[rs], %[todo]\n\t"          /* the row to update (wave-uniform) */               \
  SOLO_PGS_LANE_MASK                                                                               \
  "v_readlane_b32 s94, " SOLO_PGS64_DL_LO ", %[rs]\n\t"       /* the change of its impulse */                      \
  "v_readlane_b32 s95, " SOLO_PGS64_DL_HI ", %[rs]\n\t"                                                            \
  "v_cndmask_b32_e32 " SOLO_PGS64_LAM_LO ", " SOLO_PGS64_LAM_LO ", " SOLO_PGS64_CAND_LO ", vcc\n\t"   /* lam[row] = cand[row] and its threshold: VECTOR work behind */ \
  "v_cndmask_b32_e32 " SOLO_PGS64_LAM_HI ", " SOLO_PGS64_LAM_HI ", " SOLO_PGS64_CAND_HI ", vcc\n\t"   /* the readlanes - a scalar instruction issued there waits ~16 */ \
  "v_mul_f64 %[thr], %[tol], |" SOLO_PGS64_LAM "|\n\t"                                                       /* cycles for their SGPR writes (see above) */ \
  "s_lshl_b32 %[ri], %[rs], 1\n\t"            /* register index of the column: 2 x row */          \
  "s_set_gpr_idx_on %[ri], gpr_idx(SRC0)\n\t"                                                      \
  "s_lshl_b64 %[t], -2, %[rs]\n\t"            /* (a scalar instruction between the mode switch and the indexed VALU instruction) */ \
  "v_fma_f64 %[v], " SOLO_PGS64_COL0 ", s[94:95], %[v]\n\t"  /* v += column * change (source 0 register-indexed) */ \
  "s_set_gpr_idx_off\n\t"                                                                          \
  "s_and_b64 %[w], " PH ", %[t]\n\t"          /* the phase's rows beyond the cursor (and the wait state after the mode switch) */ \
  "v_max_f64 " SOLO_PGS64_CAND ", %[v], " SOLO_PGS64_LO "\n\t"                                                     \
  "v_min_f64 " SOLO_PGS64_CAND ", " SOLO_PGS64_CAND ", " SOLO_PGS64_HI "\n\t"                                               \
  "v_add_f64 " SOLO_PGS64_DL ", " SOLO_PGS64_CAND ", -" SOLO_PGS64_LAM "\n\t"                                              \
  "v_cmp_gt_f64_e64 %[pend], |" SOLO_PGS64_DL "|, %[thr]\n\t"                                             \
  SOLO_PGS_COUNT_ROW                                                                               \
  "s_and_b64 %[todo], %[pend], %[w]\n\t"


#define SOLO_PGS_WALK64(P, PH)                                                                     \
  ".Lpgs64_%=_" P "_row:\n\t"                                                                      \
  SOLO_PGS_ROW64(PH)                                                                               \
  "s_cbranch_scc0 .Lpgs64_%=_" P "_out\n\t"                                                        \
  SOLO_PGS_ROW64(PH)                                                                               \
  "s_cbranch_scc1 .Lpgs64_%=_" P "_row\n"                                                          \
  ".Lpgs64_%=_" P "_out:\n\t"

// friction limits = mu x the normal impulse their contact holds now.  In slot space a contact's rows are three
// consecutive SLOTS, which may straddle a 16-lane row: DPP wave_shr:1 of the two dwords (x1 = the slot below), and
// again for the slot two below (two wait states between a VALU write and the DPP read of it)
#define SOLO_PGS_LIMITS64                                                                          \
  "v_mov_b32_dpp " SOLO_PGS64_X1_LO ", " SOLO_PGS64_LAM_LO " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                 \
  "v_mov_b32_dpp " SOLO_PGS64_X1_HI ", " SOLO_PGS64_LAM_HI " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                 \
  "s_nop 0\n\t"                                                                                   \
  "v_mov_b32_dpp " SOLO_PGS64_X2_LO ", " SOLO_PGS64_X1_LO " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                \
  "v_mov_b32_dpp " SOLO_PGS64_X2_HI ", " SOLO_PGS64_X1_HI " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                \
  "v_cndmask_b32_e64 " SOLO_PGS64_X1_LO ", " SOLO_PGS64_X2_LO ", " SOLO_PGS64_X1_LO ", %[tan1]\n\t"                                                \
  "v_cndmask_b32_e64 " SOLO_PGS64_X1_HI ", " SOLO_PGS64_X2_HI ", " SOLO_PGS64_X1_HI ", %[tan1]\n\t"                                                \
  "v_mul_f64 " SOLO_PGS64_X1 ", %[mu], " SOLO_PGS64_X1 "\n\t"                                                    \
  "v_cndmask_b32_e64 " SOLO_PGS64_LO_LO ", " SOLO_PGS64_LO_LO ", " SOLO_PGS64_X1_LO ", %[tang]\n\t"    /* lo = -lim (the sign lives in the high dword) */ \
  "v_cndmask_b32_e64 " SOLO_PGS64_LO_HI ", " SOLO_PGS64_LO_HI ", -" SOLO_PGS64_X1_HI ", %[tang]\n\t"                                               \
  "v_cndmask_b32_e64 " SOLO_PGS64_HI_LO ", " SOLO_PGS64_HI_LO ", " SOLO_PGS64_X1_LO ", %[tang]\n\t"    /* hi = lim */                              \
  "v_cndmask_b32_e64 " SOLO_PGS64_HI_HI ", " SOLO_PGS64_HI_HI ", " SOLO_PGS64_X1_HI ", %[tang]\n\t"                                                \
  "v_max_f64 " SOLO_PGS64_CAND ", %[v], " SOLO_PGS64_LO "\n\t"                                                     \
  "v_min_f64 " SOLO_PGS64_CAND ", " SOLO_PGS64_CAND ", " SOLO_PGS64_HI "\n\t"                                               \
  "v_add_f64 " SOLO_PGS64_DL ", " SOLO_PGS64_CAND ", -" SOLO_PGS64_LAM "\n\t"                                              \
  "v_cmp_gt_f64_e64 %[pend], |" SOLO_PGS64_DL "|, %[thr]\n\t"

__device__ __forceinline__ int pgs_solve_gfx950(const ColumnBank<double>& A, double& v, double& lam, double& cand, double& dl,
                                                unsigned long long& pend, double& lo, double& hi, double tol, int lane, double mu,
                                                unsigned long long tan1_lanes, unsigned long long tangent_lanes,
                                                unsigned long long phase0, unsigned long long phase1, unsigned long long phase2,
                                                int iters, int& n_changed) {
  double thr, lam_o, cand_o, dl_o, lo_o, hi_o, x1, x2;
  unsigned long long w, t, todo;
  int rs, ri, it;
  asm volatile(
      "s_branch .Lpgs64_%=_entry\n\t"
      ".p2align 6\n\t"
      ".fill 6, 4, 0xbf800000\n"               // (s_nop 0: the loops at a fixed position within the 64-byte instruction lines)
      ".Lpgs64_%=_entry:\n\t"
      "s_sub_u32 %[it], 0, %[iters]\n\t"       // counts up to zero: the carry of the increment is "cap reached"
      "s_cbranch_scc0 .Lpgs64_%=_done\n"        // (no sweeps allowed)
      ".Lpgs64_%=_sweep:\n\t"
      // The FALL-THROUGH path is the one the slow robots take: in the late sweeps of a creeping robot the non-contact
      // rows (joint motors, joint limits) stand still and normal + friction rows move (91 % of the sweeps beyond the
      // fifth of the robot-steps that run to the cap; all three phases: 8 %), so a sweep that finds no work there
      // falls through both tests into the normal rows, and a sweep whose non-contact rows move takes that phase out
      // of line (p0 ... below): one taken branch per sweep - the way back - instead of two.
      "s_and_b64 %[todo], %[pend], %[ph0]\n\t"
      "s_cbranch_scc1 .Lpgs64_%=_p0\n\t"
      // ---- all normal rows, then the friction limits (a sweep that has found no work so far)
      "s_and_b64 %[todo], %[pend], %[ph1]\n\t"
      "s_cbranch_scc0 .Lpgs64_%=_a2\n"
      SOLO_PGS_WALK64("q1", "%[ph1]")
      SOLO_PGS_LIMITS64
      // ---- all friction rows
      ".Lpgs64_%=_b2:\n\t"
      "s_and_b64 %[todo], %[pend], %[ph2]\n\t"
      "s_cbranch_scc0 .Lpgs64_%=_next\n"
      ".Lpgs64_%=_p2:\n"
      SOLO_PGS_WALK64("p2", "%[ph2]")
      ".Lpgs64_%=_next:\n\t"
      "s_add_u32 %[it], %[it], 1\n\t"
      "s_cbranch_scc0 .Lpgs64_%=_sweep\n\t"       // (no carry: below the sweep cap)
      "s_branch .Lpgs64_%=_done\n"
      ".Lpgs64_%=_a2:\n\t"                        // (no work in the first two phases)
      "s_and_b64 %[todo], %[pend], %[ph2]\n\t"
      "s_cbranch_scc1 .Lpgs64_%=_p2\n\t"
      "s_branch .Lpgs64_%=_done\n"                 // (nothing pending at the start of a sweep: converged)
      // ---- out of line: the non-contact rows (joint motors, joint limits), leg by leg, then the normal rows of
      //      a sweep that has found work
      ".p2align 5\n"                            // (never executed: behind an unconditional branch)
      ".Lpgs64_%=_p0:\n"
      SOLO_PGS_WALK64("p0", "%[ph0]")
      "s_and_b64 %[todo], %[pend], %[ph1]\n\t"
      "s_cbranch_scc0 .Lpgs64_%=_b2\n"            // no normal row moves in this sweep: the friction limits stand
      SOLO_PGS_WALK64("p1", "%[ph1]")
      SOLO_PGS_LIMITS64
      "s_branch .Lpgs64_%=_b2\n"
      ".Lpgs64_%=_done:\n\t"
      : [v] "+v"(v), "={" SOLO_PGS64_LAM "}"(lam_o), "={" SOLO_PGS64_CAND "}"(cand_o), "={" SOLO_PGS64_DL "}"(dl_o), "={" SOLO_PGS64_LO "}"(lo_o), "={" SOLO_PGS64_HI "}"(hi_o),
        "=&{" SOLO_PGS64_X1 "}"(x1), "=&{" SOLO_PGS64_X2 "}"(x2), [pend] "+s"(pend), [thr] "=&v"(thr),
        [w] "=&s"(w), [t] "=&s"(t), [todo] "=&s"(todo), [rs] "=&s"(rs), [ri] "=&s"(ri), [it] "=&s"(it)
#ifdef SOLO_STAMPS
        , [nch] "+s"(n_changed)
#endif
      : "{" SOLO_PGS64_LAM "}"(lam), "{" SOLO_PGS64_CAND "}"(cand), "{" SOLO_PGS64_DL "}"(dl), "{" SOLO_PGS64_LO "}"(lo), "{" SOLO_PGS64_HI "}"(hi),
        SOLO_PGS_LANE_OPERAND(lane) [tol] "v"(tol), [mu] "v"(mu), [iters] "s"(iters), [ph0] "s"(phase0), [ph1] "s"(phase1), [ph2] "s"(phase2),
        [tan1] "s"(tan1_lanes), [tang] "s"(tangent_lanes),
        "{" SOLO_PGS64_BANK0 "}"(A.a0), "{" SOLO_PGS64_BANK1 "}"(A.a1)
      : "vcc", "scc", "s94", "s95");
  lam = lam_o; cand = cand_o; dl = dl_o; lo = lo_o; hi = hi_o;
  (void)n_changed; (void)x1; (void)x2;
  return it + iters;
}

#undef SOLO_PGS_LIMITS64
#undef SOLO_PGS_WALK64
#undef SOLO_PGS_ROW64
#undef SOLO_PGS_COUNT_ROW

}  // namespace solo
