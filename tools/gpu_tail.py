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
n = 4096
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
    span = t1.max() - t0.min()
    print('S=%d streams=%d: rollout %.3f ms = %.2f us/step ; ticks: span %d (%.1f MHz tick rate) ; per-robot life/step pct 1/50/90/99/max = %s ; mean %.0f ; mean/span = %.2f ; start skew %d' % (
      spl, streams, ms, ms * 1e3 / spl, span, span / (ms * 1e3), (np.percentile(life, [1, 50, 90, 99, 100]) / spl).astype(int).tolist(),
      life.mean() / spl, life.mean() / span, t0.max() - t0.min()))
    for g_ in range(streams):
      lo, hi = n * g_ // streams, n * (g_ + 1) // streams
      print('    slice %d: first start %d last end %d (rel. to global first start)' % (g_, t0[lo:hi].min() - t0.min(), t1[lo:hi].max() - t0.min()))
    env._close()
