"""MEASUREMENT: is a K-step launch bound by its SLOWEST ROBOT (K sequential steps of one wave: a critical path no
scheduling shortens) or by the wave slots (work / slots)?  From the first and last s_memtime stamp of every robot's wave
(make -C gym_solo_amd/csrc stamps_light; one robot per wave: run it with migrate_steps = 0).
  python tools/gpu_critical_path.py float64 20 4096"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('SOLO_HIP_LIB', os.path.join(ROOT, 'gym_solo_amd', 'csrc', 'libsolo_hip_stamps_light.so'))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from gym_solo_amd import abi

dtype = sys.argv[1] if len(sys.argv) > 1 else 'float64'
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tdt = torch.float32 if dtype == 'float32' else torch.float64
slots = 1024 * 4   # (four waves per SIMD in both precisions since round 5; the light stamps build keeps the product's residency)
for n in [int(x) for x in sys.argv[3:]] or [4096]:
  env = bench.build_env(n, 0, dtype, steps_per_launch=k, rollout_streams=1, migrate_steps=0)
  eng = env.engine
  gen = torch.Generator(device='cuda').manual_seed(1234)
  bench.desynchronise_episodes(eng, gen)
  pool = lambda steps: (torch.rand(steps, n, abi.NUM_JOINTS, device='cuda', dtype=tdt, generator=gen) * 2 - 1) * 6.283185307179586
  eng.lib.solo_engine_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
  for rep in range(3):
    eng.rollout(pool(k), abi.STEP_ALL)
    torch.cuda.synchronize()
    buf = np.zeros((n, 32), dtype=np.uint64)
    assert eng.lib.solo_engine_debug_stamps(eng._h, buf.ctypes.data, 1 if dtype == 'float32' else 0) == 0
    t0 = buf[:, 0].astype(np.int64); t1 = buf[:, 14].astype(np.int64)
    life = t1 - t0
    span = t1.max() - t0.min()   # (ticks of the 100-MHz real-time counter: 10 ns)
    late = (t0 - t0.min()) > 0.02 * span
    worst = np.argsort(t1)[-5:]
    print('%s N = %d K = %d: launch span %.1f us; a robot\'s wave lives %.2f (mean) / %.2f (p90) / %.2f (p99) / %.2f (max) of the span; work / slots = %.2f of the span; '
          '%d waves had to wait for a slot (their lives: %.2f of the span on average)'
          % (dtype, n, k, span / 100.0, life.mean() / span, np.percentile(life, 90) / span, np.percentile(life, 99) / span, life.max() / span, life.sum() / slots / span,
             late.sum(), life[late].mean() / span if late.any() else 0), flush=True)
    print('   the five waves that end last: started at %s of the span, lived %s of it' % (
      np.round((t0[worst] - t0.min()) / span, 2).tolist(), np.round(life[worst] / span, 2).tolist()), flush=True)
  env._close()
