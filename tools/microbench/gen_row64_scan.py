"""MICROBENCH generator (not product): where does the f64 Gauss-Seidel row update of solo_pgs_gfx950.h WAIT?
Writes row64_scan.hip: the row as a list of instructions, one kernel per variant -
  * the row as it is / candidate orders of the same instructions,
  * each of them with ONE `s_nop 0` inserted in front of instruction i (i = 0 .. n): where the insertion costs nothing,
    the wave was waiting anyway (a slot that independent work could fill); where it costs 4 cycles, the issue is the bound.
One wave per SIMD, 16 straight-line repetitions per loop iteration, the empty loop subtracted (method of simd_rate.hip).
  python tools/microbench/gen_row64_scan.py > tools/microbench/row64_scan.hip
  hipcc -O3 --offload-arch=gfx950 -o tools/microbench/row64_scan tools/microbench/row64_scan.hip"""
import sys

I = dict(
  ff1='s_ff1_i32_b64 %[rs], %[todo]',
  eq='v_cmp_eq_u32_e32 vcc, %[rs], %[lane]',
  rl0='v_readlane_b32 s94, v54, %[rs]',
  rl1='v_readlane_b32 s95, v55, %[rs]',
  ri='s_lshl_b32 %[ri], %[rs], 1',
  on='s_set_gpr_idx_on %[ri], 1',
  t='s_lshl_b64 %[t], -2, %[rs]',
  fma='v_fma_f64 %[v], v[64:65], s[94:95], %[v]',
  off='s_set_gpr_idx_off',
  w='s_and_b64 %[w], %[ph], %[t]',
  c0='v_cndmask_b32_e32 v50, v50, v52, vcc',
  c1='v_cndmask_b32_e32 v51, v51, v53, vcc',
  max='v_max_f64 v[52:53], %[v], %[lo]',
  min='v_min_f64 v[52:53], v[52:53], %[hi]',
  # (the ping-pong forms: the new candidates in a second register pair, so that lam[row] = cand[row] can follow the compare)
  maxn='v_max_f64 v[56:57], %[v], %[lo]',
  minn='v_min_f64 v[56:57], v[56:57], %[hi]',
  addn='v_add_f64 v[54:55], v[56:57], -v[50:51]',
  thr='v_mul_f64 %[thr], %[tol], |v[50:51]|',
  add='v_add_f64 v[54:55], v[52:53], -v[50:51]',
  cmp='v_cmp_gt_f64_e64 %[pend], |v[54:55]|, %[thr]',
  sm='s_lshl_b64 vcc, 1, %[rs]',
  cmpv='v_cmp_gt_f64_e64 vcc, |v[54:55]|, %[thr]',
  todov='s_and_b64 %[todo], vcc, %[w]',
  cmpx='v_cmpx_gt_f64_e64 |v[54:55]|, %[thr]',
  todox='s_and_b64 %[todo], exec, %[w]',
  exon='s_mov_b64 exec, -1',
  br='s_nop 0',   # (the branch's issue slot)
  todo='s_and_b64 %[todo], %[pend], %[w]',
  keep='s_or_b64 %[todo], %[todo], %[ph]',   # keeps the walk going (not an instruction of the product's row)
  nop='s_nop 0',
  vnop='v_nop',
  snop='s_nop 0',
)

ORDERS = {
  'product': 'ff1 eq rl0 rl1 ri on t fma off w c0 c1 max min thr add cmp br todo keep',
  # lam[row] = cand[row] and thr behind the compare (new candidates in a second pair; a real loop alternates the pairs)
  'deferred': 'ff1 rl0 rl1 ri on t fma off w maxn minn addn cmp eq c0 c1 thr br todo keep',
  'round5': 'ff1 sm rl0 rl1 c0 c1 thr ri on t fma off w max min add cmp todo br keep',
  'round5_vcc': 'ff1 sm rl0 rl1 c0 c1 thr ri on t fma off w max min add cmpv todov br keep',
  'round5_nobr': 'ff1 sm rl0 rl1 c0 c1 thr ri on t fma off w max min add cmp todo keep',
  'round5_w_late': 'ff1 sm rl0 rl1 c0 c1 thr ri on t fma off br max min add cmp w todo keep',
  'fillA': 'ff1 rl0 rl1 eq c0 c1 thr ri on t fma off w max min add cmp br todo keep',
  'fillA3': 'ff1 rl0 rl1 eq c0 c1 ri on t fma off w thr max min add cmp br todo keep',
  'eq_first': 'ff1 eq rl0 rl1 c0 c1 thr ri on t fma off w max min add cmp br todo keep',
  'salu_first': 'ff1 ri t w rl0 rl1 eq c0 c1 thr on snop fma off snop max min add cmp br todo keep',
  'def_fillA': 'ff1 rl0 rl1 eq c0 c1 thr ri on t fma off w maxn minn addn cmp br todo keep',
  'def_3A_1B': 'ff1 rl0 rl1 eq c0 c1 ri on t fma off w maxn minn addn cmp thr br todo keep',
  'def_1A_3B': 'ff1 rl0 rl1 thr ri on t fma off w maxn minn addn cmp eq c0 c1 br todo keep',
  'def_2A_2B': 'ff1 rl0 rl1 eq c0 ri on t fma off w maxn minn addn cmp c1 thr br todo keep',
  'deferred_b': 'ff1 rl0 rl1 ri on t fma off w maxn minn addn cmp eq br todo keep c0 c1 thr',
  'deferred_c': 'ff1 rl0 rl1 ri on t fma off maxn minn addn cmp w eq c0 c1 thr br todo keep',
}


def kernel(name, seq):
  body = ''.join('                 "%s\\n"\n' % I[k] for k in seq)
  return '''template <> __device__ __forceinline__ void body<%(name)s>(double& v, double& lam, double& cand, double& dl, double& c2, double& acc, double b, double c, int lane) {
  unsigned long long pend = 0, w = 0, t = 0, todo = 0x0000000009240924ull;
  const unsigned long long ph = 0x0000000009240924ull;
  int rs = 0, ri = 0;
  double thr = c;
  const double lo = -b, hi = b, tol = c;
  typedef double d16 __attribute__((ext_vector_type(16)));
  const d16 z0 = b * 1e-3, z1 = c;
  asm volatile(".rept 16\\n"
%(body)s                 ".endr\\n"
               : [v] "+v"(v), "+{v[50:51]}"(lam), "+{v[52:53]}"(cand), "+{v[54:55]}"(dl), "+{v[56:57]}"(c2), [thr] "+v"(thr), [pend] "+s"(pend), [w] "+s"(w),
                 [t] "+s"(t), [todo] "+s"(todo), [rs] "+s"(rs), [ri] "+s"(ri)
               : [lane] "v"(lane), [tol] "v"(tol), [lo] "v"(lo), [hi] "v"(hi), [ph] "s"(ph), "{v[64:95]}"(z0), "{v[96:127]}"(z1)
               : "vcc", "scc", "s94", "s95");
  acc += (double)todo;
}
''' % dict(name=name, body=body)


variants = []   # (enum name, label, sequence)
for oname, order in ORDERS.items():
  seq = order.split()
  variants.append(('k_%s' % oname, '%s: %s' % (oname, ' '.join(seq)), seq))
  if oname == 'product':
    ia, ib = seq.index('rl1') + 1, seq.index('cmp') + 1
    for k in range(1, 11):
      variants.append(('k_vnopA%d' % k, '  product + %d v_nop behind the readlanes' % k, seq[:ia] + ['vnop'] * k + seq[ia:]))
    for k in range(1, 11):
      variants.append(('k_vnopB%d' % k, '  product + %d v_nop behind the compare' % k, seq[:ib] + ['vnop'] * k + seq[ib:]))
    for k in range(1, 9):
      variants.append(('k_vnopAB%d' % k, '  product + %d v_nop behind the readlanes AND behind the compare' % k, seq[:ia] + ['vnop'] * k + seq[ia:ib] + ['vnop'] * k + seq[ib:]))
  if False:
    for i in range(len(seq) + 1):
      s2 = seq[:i] + ['nop'] + seq[i:]
      variants.append(('k_%s_nop%d' % (oname, i), '  %s + s_nop in front of #%d (%s)' % (oname, i, seq[i] if i < len(seq) else 'end'), s2))
    for i in range(len(seq) + 1):
      s2 = seq[:i] + ['nop', 'nop', 'nop'] + seq[i:]
      variants.append(('k_%s_3nop%d' % (oname, i), '  %s + 3 s_nop in front of #%d (%s)' % (oname, i, seq[i] if i < len(seq) else 'end'), s2))

out = ['// GENERATED by gen_row64_scan.py (MICROBENCH, not product) - see that file', '#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <vector>', '#include <algorithm>',
       'enum Mode { k_empty, %s, kModes };' % ', '.join(v[0] for v in variants),
       'template <int MODE> __device__ __forceinline__ void body(double& v, double& lam, double& cand, double& dl, double& c2, double& acc, double b, double c, int lane) {}']
for name, label, seq in variants:
  out.append(kernel(name, seq))
out.append('''template <int MODE>
__global__ __launch_bounds__(64, 4) void k(double* out, unsigned long long* ticks, int iters) {
  double v = threadIdx.x * 1e-3, lam = 1, cand = 2, dl = 3e-3, c2 = 4, acc = 5;
  const double b = 1.0001, c = 1e-4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) body<MODE>(v, lam, cand, dl, c2, acc, b, c, (int)threadIdx.x);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + threadIdx.x] = v + lam + cand + dl + c2 + acc;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int MODE> double run(int blocks, int iters) {
  double* out; unsigned long long* t;
  (void)hipMalloc(&out, blocks * 64 * 8); (void)hipMalloc(&t, blocks * 8);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, t, iters); (void)hipDeviceSynchronize(); }
  std::vector<unsigned long long> h(blocks); (void)hipMemcpy(h.data(), t, blocks * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  (void)hipFree(out); (void)hipFree(t);
  return (double)h[blocks / 2];
}
template <int MODE> void report(const char* label, int n, double empty) {
  const double t = run<MODE>(1024, 400) - empty;
  printf("%7.1f cycles per row (%d instr)  %s\\n", t / (400 * 16.0), n, label);
}
int main() {
  const double e = run<k_empty>(1024, 400);
  printf("one wave per SIMD; empty loop %.1f ticks per iteration, subtracted\\n", e / 400);''')
for name, label, seq in variants:
  out.append('  report<%s>("%s", %d, e);' % (name, label.replace('%', '%%'), len(seq)))
out.append('  return 0;\n}')
print('\n'.join(out))
