"""DIAGNOSTIC (make -C gym_solo_amd/csrc stamps): where does a SINGLE-STEP launch (the closed loop's granularity) spend its
time - per phase, for the median wave and for the 40 longest-lived waves (the ones the launch waits for)?  In-kernel
s_memtime stamps (shader clock cycles); the stamps build holds 14 workgroups per CU, not 16:
shares, not run times.
  DTYPE=float64 python tools/gpu_step_phases.py [N]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('SOLO_HIP_LIB', os.path.join(ROOT, 'gym_solo_amd', 'csrc', 'libsolo_hip_stamps.so'))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from gym_solo_amd import abi
names = ['loads+sync', 'kinematics', 'leg inertia', 'bias forces', 'leg sum', 'chol+solve', 'rows', 'column build', 'Gauss-Seidel', 'finish+nan check', 'term+record',
         'outputs in place', 'restart+done', 'state store']
DTYPE = os.environ.get('DTYPE', 'float64')
TD = torch.float32 if DTYPE == 'float32' else torch.float64
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = bench.build_env(n, 0, DTYPE, steps_per_launch=1, rollout_streams=1, migrate_steps=0)
eng = env.engine
g = torch.Generator(device='cuda').manual_seed(1234)
bench.desynchronise_episodes(eng, g)
eng.lib.solo_engine_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
acc_med, acc_slow, spans = [], [], []
for rep in range(10):
  a = (torch.rand(n, 12, device='cuda', dtype=TD, generator=g) * 2 - 1) * (2 * np.pi)
  eng.step(a, abi.STEP_ALL)
  torch.cuda.synchronize()
  buf = np.zeros((n, 32), dtype=np.uint64)
  assert eng.lib.solo_engine_debug_stamps(eng._h, buf.ctypes.data, 1 if DTYPE == 'float32' else 0) == 0
  st = buf[:, :15].astype(np.int64)
  d = np.diff(st, axis=1)
  life = st[:, 14] - st[:, 0]
  slow = np.argsort(life)[-40:]
  acc_med.append(np.median(d, axis=0)); acc_slow.append(d[slow].mean(axis=0)); spans.append((life.max(), np.median(life), life[slow].mean()))
  its = (buf[:, 15] & 0xffff).astype(np.int64)
med, slw = np.mean(acc_med, axis=0), np.mean(acc_slow, axis=0)
print('%s, N = %d, single-step launches (10): wave life median %.0f ticks, the 40 longest-lived %.0f, the longest %.0f (shader cycles)' % (
  DTYPE, n, np.mean([s[1] for s in spans]), np.mean([s[2] for s in spans]), np.mean([s[0] for s in spans])))
for k, nm in enumerate(names):
  print('   %-18s median wave %7.0f (%4.1f %%)    40 longest-lived %7.0f (%4.1f %%)' % (nm, med[k], 100 * med[k] / med.sum(), slw[k], 100 * slw[k] / slw.sum()))
env._close()
