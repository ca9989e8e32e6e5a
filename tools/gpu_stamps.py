"""DIAGNOSTIC: per-phase cycle shares of the step kernel from in-kernel s_memtime stamps
(libsolo_hip_stamps.so; never quote this build's run time, only its shares)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('SOLO_HIP_LIB', os.path.join(ROOT, 'gym_solo_amd', 'csrc', 'libsolo_hip_stamps.so'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from gym_solo_amd import abi
from bench import build_env
names = ['loads+sync', 'kinematics', 'crba', 'rne bias', 'schur+sum', 'chol+solve', 'rows', 'A build', 'PGS', 'finish+nan check', 'term+record', 'restart+done', 'loop exit', 'epilogue']
if 'epilogue' in os.environ['SOLO_HIP_LIB']:  # (make stamps_epilogue: stamps 1 .. 12 sit inside the output epilogue; per-step figures = per launch / steps)
  names = ['record fence', 'event + done', 'euler angles', 'observations', 'rewards', 'sync', 'returns'] + ['-'] * 5 + ['the steps', 'tail']
DTYPE = os.environ.get('DTYPE', 'float32')
TD = torch.float32 if DTYPE == 'float32' else torch.float64
for n in ([] if 'epilogue' in os.environ['SOLO_HIP_LIB'] else [int(a) for a in sys.argv[1:]] or (1024, 4096)):
  env = build_env(n, 0, DTYPE)
  eng = env.engine
  g = torch.Generator(device='cuda').manual_seed(1234)
  acts = (torch.rand(64, n, 12, device='cuda', dtype=TD, generator=g) * 2 - 1) * (2 * np.pi)
  eng.rollout(acts, abi.STEP_ALL)
  eng.step(acts[0], abi.STEP_ALL)
  buf = np.zeros((n, 32), dtype=np.uint64)
  eng.lib.solo_engine_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
  assert eng.lib.solo_engine_debug_stamps(eng._h, buf.ctypes.data, 1 if DTYPE == 'float32' else 0) == 0
  d = np.diff(buf[:, :15].astype(np.int64), axis=1)
  med = np.median(d, axis=0)
  tot = np.median(buf[:, 14].astype(np.int64) - buf[:, 0].astype(np.int64))
  span = (buf[:, 14].max() - buf[:, 0].min())
  print(f'N={n}: median wave lifetime {tot} ticks, whole-grid span {span} ticks')
  for k, nm in enumerate(names):
    print(f'   {nm:14s} {med[k]:9.0f}  {100*med[k]/tot:5.1f}%')
  t0 = buf[:, 0].astype(np.int64); t1 = buf[:, 14].astype(np.int64)
  life = t1 - t0
  print('   wave life percentiles 50/90/99/max: %s ; start skew (last start - first start) %d ; first start -> last end %d ticks' % (np.percentile(life, [50, 90, 99, 100]).astype(int).tolist(), t0.max() - t0.min(), t1.max() - t0.min()))
  its = (buf[:, 15] & 0xffff).astype(np.int64); ncs = ((buf[:, 15] >> 16) & 0xff).astype(np.int64)
  print('   sweeps executed: mean %.1f  hist(0,5,10,20,30,40,49,50)=%s' % (its.mean(), np.histogram(its, bins=[0,5,10,20,30,40,49,50,51])[0].tolist()))
  for c in range(0, 8):
    m = ncs == c
    if m.any(): print('   nc=%d: %4d robots, mean sweeps %.1f, at cap %.0f%%, mean wave life %.0f' % (c, m.sum(), its[m].mean(), 100*(its[m]>=49).mean(), (buf[m,14].astype(np.int64)-buf[m,0].astype(np.int64)).mean()))
  env._close()

# ---- fused launches (the bench configuration): how unequal are the robots' 100-step totals? ----
for n, spl in [tuple(int(x) for x in f.split(':')) for f in os.environ.get('FUSED', '1024:100,4096:100').split(',')]:
  env = build_env(n, 0, DTYPE, steps_per_launch=spl, rollout_streams=1)
  eng = env.engine
  g = torch.Generator(device='cuda').manual_seed(99)
  acts = (torch.rand(500, n, 12, device='cuda', dtype=TD, generator=g) * 2 - 1) * (2 * np.pi)
  eng.rollout(acts, abi.STEP_ALL)                      # into the flailing steady state
  eng.rollout(acts[:spl], abi.STEP_ALL)                # ONE fused launch: its stamps are read back
  buf = np.zeros((n, 32), dtype=np.uint64)
  eng.lib.solo_engine_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
  assert eng.lib.solo_engine_debug_stamps(eng._h, buf.ctypes.data, 1 if DTYPE == 'float32' else 0) == 0
  t0 = buf[:, 0].astype(np.int64); t1 = buf[:, 14].astype(np.int64)
  per_step = (t1 - t0) / spl
  acc = buf[:, 16:32].astype(np.int64)   # acc[:, i] = ticks before stamp i, summed over the launch; acc[:, 15] = sweeps
  order = np.argsort(per_step)
  groups = (('fastest 10%', order[:n // 10]), ('middle 10%', order[n * 45 // 100:n * 55 // 100]), ('slowest 1%', order[-max(1, n // 100):]))
  print('   per-step ticks by phase (launch totals / steps), robots grouped by their launch total:')
  print('   %-14s' % 'phase' + ''.join('%14s' % g[0] for g in groups))
  for k, nm in enumerate(names):
    print('   %-14s' % nm + ''.join('%14.0f' % (acc[idx, k + 1].mean() / spl) for _, idx in groups))
  print('   %-14s' % 'changed rows' + ''.join('%14.2f' % (acc[idx, 0].mean() / spl) for _, idx in groups))
  print('   %-14s' % 'sweeps/step' + ''.join('%14.2f' % (acc[idx, 15].mean() / spl) for _, idx in groups))
  print('fused N=%d S=%d: per-robot mean step (ticks) percentiles 1/50/90/99/max: %s ; launch makespan/steps = %.0f ; mean %.0f' % (
    n, spl, np.percentile(per_step, [1, 50, 90, 99, 100]).astype(int).tolist(), (t1.max() - t0.min()) / spl, per_step.mean()))
  env._close()

if DTYPE != 'float32': sys.exit(0)
# ---- is a robot's cost persistent from one fused launch to the next? (would cost-sorted slices pay?)
n, spl = 4096, 100
env = build_env(n, 0, 'float32', steps_per_launch=spl, rollout_streams=1)
eng = env.engine
g = torch.Generator(device='cuda').manual_seed(7)
acts = (torch.rand(400, n, 12, device='cuda', dtype=torch.float32, generator=g) * 2 - 1) * (2 * np.pi)
eng.rollout(acts, abi.STEP_ALL)
tot = []
for j in range(4):
  eng.rollout(acts[j * spl:(j + 1) * spl], abi.STEP_ALL)
  buf = np.zeros((n, 32), dtype=np.uint64)
  assert eng.lib.solo_engine_debug_stamps(eng._h, buf.ctypes.data, 1 if DTYPE == 'float32' else 0) == 0
  tot.append((buf[:, 14].astype(np.int64) - buf[:, 0].astype(np.int64)) / spl)
  hw = (buf[:, 15] >> 28).astype(np.int64); xcc = ((buf[:, 15] >> 24) & 0xf).astype(np.int64)
  simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
  placements = globals().setdefault('placements', [])
  placements.append(((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd))
tot = np.array(tot)
# placement: which SIMD ran each robot's wave (XCC, SE, SH, CU, SIMD from HW_ID), is it the same every launch?
place = []
print('launch-to-launch correlation of per-robot cost (100-step launches): %s' % np.round([np.corrcoef(tot[j], tot[j + 1])[0, 1] for j in range(3)], 3).tolist())
pl = np.array(placements)
print('distinct SIMD ids seen: %d ; waves per SIMD min/max: %s ; same placement as previous launch: %s' % (
  len(np.unique(pl[0])), np.bincount(np.unique(pl[0], return_inverse=True)[1]).min().__repr__() + '/' + np.bincount(np.unique(pl[0], return_inverse=True)[1]).max().__repr__(),
  [float((pl[j] == pl[j + 1]).mean()) for j in range(3)]))
u, inv = np.unique(pl[1], return_inverse=True)
simd_sum = np.bincount(inv, weights=tot[1]); cnt = np.bincount(inv)
print('per-SIMD sum of its robots cost: mean %.0f max %.0f (x%.2f) ; per-robot cost: mean %.0f max %.0f ; block ids sharing SIMD of block 0: %s' % (
  simd_sum.mean(), simd_sum.max(), simd_sum.max() / simd_sum.mean(), tot[1].mean(), tot[1].max(), np.where(pl[1] == pl[1][0])[0].tolist()))
for j in range(3):
  order = np.argsort(tot[j])  # sort by the PREVIOUS launch's cost, look at this launch's maxima per quartile
  q = [tot[j + 1][order[i * n // 4:(i + 1) * n // 4]].max() for i in range(4)]
  print('   quartiles by previous cost -> max cost now: %s ; overall max %d mean %d' % (np.array(q).astype(int).tolist(), tot[j + 1].max(), tot[j + 1].mean()))
env._close()
