"""DIAGNOSTIC (CPU): replay the slow robot-steps collected by tools/gpu_dump_slow.py on the wave
emulator's trace build and look at what the Gauss-Seidel iteration does sweep by sweep."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from gym_solo_amd import abi
from helpers import make_abi
import emu_kernel
DTYPE = os.environ.get('DTYPE', 'float32')
d = np.load(os.path.join(ROOT, 'gpurun_out', 'slow_steps_%s.npz' % DTYPE))
ca, ma = make_abi(DTYPE)
e = emu_kernel.EmuEngine(ca, ma, 1, variant='trace')
lib = e.lib
lib.solo_emu_trace.restype = C.c_int
lam = np.zeros((64, 64)); v = np.zeros((64, 64)); pend = np.zeros(64, dtype=np.uint64)
sel = [i for i in range(len(d['sweeps'])) if d['sweeps'][i] >= 50][:int(sys.argv[1]) if len(sys.argv) > 1 else 12]
for i in sel:
  e.state[0] = d['state'][i]
  e.step(d['action'][i][None], abi.STEP_PHYSICS)
  ns = lib.solo_emu_trace(lam.ctypes.data_as(C.POINTER(C.c_double)), v.ctypes.data_as(C.POINTER(C.c_double)),
                          pend.ctypes.data_as(C.POINTER(C.c_ulonglong)))
  L = lam[:ns]
  live = np.where(np.abs(L).max(axis=0) > 0)[0]
  # exact periodicity of the per-sweep state?
  per = None
  for p in (1, 2, 3, 4, 6, 8):
    if ns > 2 * p + 2 and np.array_equal(L[ns - 1], L[ns - 1 - p]) and np.array_equal(v[ns - 1], v[ns - 1 - p]) and np.array_equal(L[ns - 2], L[ns - 2 - p]):
      per = p; break
  first = None
  if per:
    for s0 in range(ns - per):
      if all(np.array_equal(L[s], L[s + per]) and np.array_equal(v[s], v[s + per]) for s in range(s0, ns - per)):
        first = s0; break
  dl = np.abs(np.diff(L, axis=0))
  rel = dl.max(axis=1) / (np.abs(L).max() + 1e-30)
  nch = (dl > 0).sum(axis=1)
  print('case %d: gpu sweeps %d, emu sweeps %d, nc %d, live rows %d ; exact period %s from sweep %s ; rows changing per sweep (last 10) %s ; max rel change per sweep: s5 %.1e s10 %.1e s20 %.1e s30 %.1e s48 %.1e' % (
    i, d['sweeps'][i], ns, d['nc'][i], len(live), per, first, nch[-10:].tolist(), *(rel[min(k, len(rel) - 1)] for k in (5, 10, 20, 30, 47))))
if len(sys.argv) > 2:
  i = int(sys.argv[2])
  e.state[0] = d['state'][i]
  e.step(d['action'][i][None], abi.STEP_PHYSICS)
  ns = lib.solo_emu_trace(lam.ctypes.data_as(C.POINTER(C.c_double)), v.ctypes.data_as(C.POINTER(C.c_double)),
                          pend.ctypes.data_as(C.POINTER(C.c_ulonglong)))
  L = lam[:ns]
  ch = np.where((np.diff(L[-12:], axis=0) != 0).any(axis=0))[0]
  print('changing lanes', ch.tolist())
  np.set_printoptions(precision=9, linewidth=200)
  for s in range(ns - 10, ns):
    print(s, 'lam', np.array([np.float32(x) for x in L[s, ch]]), 'v', np.array([np.float32(x) for x in v[s, ch]]), 'pend %x' % pend[s])
