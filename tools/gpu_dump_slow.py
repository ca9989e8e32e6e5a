"""DIAGNOSTIC: collect (state before, action) of robot-steps whose Gauss-Seidel ran many sweeps
(heavy stamp build, single-step launches), for replay on the CPU emulator.  DTYPE=float64 for the f64 kernels."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('SOLO_HIP_LIB', os.path.join(ROOT, 'gym_solo_amd', 'csrc', 'libsolo_hip_stamps.so'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from gym_solo_amd import abi
from bench import build_env
n = 4096
DTYPE = os.environ.get('DTYPE', 'float32')
TD = torch.float32 if DTYPE == 'float32' else torch.float64
env = build_env(n, 0, DTYPE)
eng = env.engine
g = torch.Generator(device='cuda').manual_seed(1234)
acts = (torch.rand(400, n, 12, device='cuda', dtype=TD, generator=g) * 2 - 1) * (2 * np.pi)
eng.lib.solo_engine_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
buf = np.zeros((n, 32), dtype=np.uint64)
states, actions, sweeps, ncs = [], [], [], []
hist = np.zeros(52, dtype=np.int64)
for k in range(400):
  before = eng.state.clone()
  eng.step(acts[k], abi.STEP_ALL)
  assert eng.lib.solo_engine_debug_stamps(eng._h, buf.ctypes.data, 1 if DTYPE == 'float32' else 0) == 0
  its = (buf[:, 15] & 0xffff).astype(np.int64)
  nc = ((buf[:, 15] >> 16) & 0xff).astype(np.int64)
  hist += np.bincount(np.minimum(its, 51), minlength=52)
  if k < 100:
    continue
  idx = np.where(its >= 25)[0]
  if len(idx):
    b = before.cpu().numpy(); a = acts[k].cpu().numpy()
    for i in idx[:8]:
      states.append(b[i]); actions.append(a[i]); sweeps.append(its[i]); ncs.append(nc[i])
print('sweep histogram over all robot-steps:', hist.tolist())
print('collected', len(states), 'slow robot-steps; sweeps', np.bincount(np.array(sweeps))[25:].tolist())
np.savez(os.path.join(ROOT, 'gpurun_out', 'slow_steps_%s.npz' % DTYPE), state=np.array(states), action=np.array(actions), sweeps=np.array(sweeps), nc=np.array(ncs))
