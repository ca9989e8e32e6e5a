"""DIAGNOSTIC: how unequal are the robots of one fused launch, and how much of the launch is tail?
Uses the light stamp build (make -C gym_solo_amd/csrc stamps_light): one s_memtime at each wave's
start and end, nothing per step.  usage: gpu_tail.py [steps_per_launch ...]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('SOLO_HIP_LIB', os.path.join(ROOT, 'gym_solo_amd', 'csrc', 'libsolo_hip_stamps_light.so'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from gym_solo_amd import abi
from bench import build_env
n = int(os.environ.get('N', '4096'))
for spl in [int(a) for a in sys.argv[1:]] or (250, 20, 1):
  for streams in (1, 2):
    env = build_env(n, 0, 'float32', steps_per_launch=spl, rollout_streams=streams)
    eng = env.engine
    g = torch.Generator(device='cuda').manual_seed(99)
    acts = (torch.rand(500 + spl, n, 12, device='cuda', dtype=torch.float32, generator=g) * 2 - 1) * (2 * np.pi)
    eng.rollout(acts[:500], abi.STEP_ALL)                      # into the flailing steady state
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.rollout(acts[500:], abi.STEP_ALL)                      # ONE fused launch per slice
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    buf = np.zeros((n, 32), dtype=np.uint64)
    eng.lib.solo_engine_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert eng.lib.solo_engine_debug_stamps(eng._h, buf.ctypes.data, 1) == 0
    t0 = buf[:, 0].astype(np.int64); t1 = buf[:, 14].astype(np.int64)
    life = (t1 - t0)
    # (s_memtime counters are not synchronised across CUs: only per-wave DIFFERENCES are meaningful)
    print('S=%d streams=%d N=%d: rollout %.3f ms = %.2f us/step ; per-robot wave lifetime per step (ticks) pct 1/50/90/99/max = %s ; mean %.0f ; max/mean %.2f' % (
      spl, streams, n, ms, ms * 1e3 / spl, (np.percentile(life, [1, 50, 90, 99, 100]) / spl).astype(int).tolist(),
      life.mean() / spl, life.max() / life.mean()))
    hw = (buf[:, 15] >> 28).astype(np.int64); xcc = ((buf[:, 15] >> 24) & 0xf).astype(np.int64)
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    print('    by XCC: ' + ' '.join('%d:%d/%.0f' % (x, (xcc == x).sum(), life[xcc == x].mean() / spl) for x in np.unique(xcc)))
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    u, inv = np.unique(cuid, return_inverse=True)
    cnt = np.bincount(inv); cmean = np.bincount(inv, weights=life / spl) / cnt
    print('    CUs used %d ; waves per CU min/max %d/%d ; per-CU mean life: min %.0f p50 %.0f max %.0f ; corr(waves on CU, mean life) %.2f' % (
      len(u), cnt.min(), cnt.max(), cmean.min(), np.median(cmean), cmean.max(), np.corrcoef(cnt, cmean)[0, 1] if cnt.std() > 0 else 0))
    sid = cuid * 4 + simd
    u2, inv2 = np.unique(sid, return_inverse=True)
    cnt2 = np.bincount(inv2)
    print('    SIMDs used %d ; waves per SIMD histogram %s ; mean life by waves-on-SIMD: %s' % (
      len(u2), np.bincount(cnt2).tolist(), {int(c): int((life / spl)[cnt2[inv2] == c].mean()) for c in np.unique(cnt2)}))
    env._close()
