"""MEASUREMENT (diagnostic stamps build): the Gauss-Seidel work per robot-step in the benchmark workload's steady state -
sweeps per robot-step, and for the robot-steps that run to the sweep cap how many rows move per sweep - from single-step
launches of libsolo_hip_stamps.so (per launch: every robot's sweep count and its number of row updates).
  make -C gym_solo_amd/csrc stamps ; DTYPE=float64 python tools/gpu_sweep_histogram.py [steps] > profiles/round4_sweep_histogram_f64.log"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('SOLO_HIP_LIB', os.path.join(ROOT, 'gym_solo_amd', 'csrc', 'libsolo_hip_stamps.so'))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gym_solo_amd import abi
import bench
DTYPE = os.environ.get('DTYPE', 'float64')
RESID = float(os.environ.get('RESID', '0'))
WARM = int(os.environ.get('WARM', '0'))
TD = torch.float32 if DTYPE == 'float32' else torch.float64
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = 4096
from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig, Solo8VanillaEnv
from gym_solo_amd.workloads import register_benchmark_workload
cfg = Solo8VanillaConfig()
cfg.num_envs, cfg.dtype, cfg.auto_reset, cfg.steps_per_launch = n, DTYPE, True, 100
cfg.solver_residual_threshold = RESID
if WARM:
  cfg.solver_warm_start = True
env = Solo8VanillaEnv(config=cfg, copy_outputs=False)
register_benchmark_workload(env, max_steps=1000)
env._ensure_program()
eng = env.engine
g = torch.Generator(device='cuda').manual_seed(1234)
bench.desynchronise_episodes(eng, g)
eng.lib.solo_engine_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
sweeps, rows, spheres = [], [], []
buf = np.zeros((n, 32), dtype=np.uint64)
for k in range(steps):
  a = (torch.rand(n, 12, device='cuda', dtype=TD, generator=g) * 2 - 1) * (2 * np.pi)
  eng.step(a, abi.STEP_ALL)
  assert eng.lib.solo_engine_debug_stamps(eng._h, buf.ctypes.data, 1 if DTYPE == 'float32' else 0) == 0
  sweeps.append((buf[:, 15] & 0xffff).astype(np.int64)); spheres.append(((buf[:, 15] >> 16) & 0xff).astype(np.int64))
  rows.append((buf[:, 15] >> 48).astype(np.int64))      # row updates of the step
sweeps, rows, spheres = np.concatenate(sweeps), np.concatenate(rows), np.concatenate(spheres)
cap = int(eng.cfg.solver_iterations)
print('%s, %d robots x %d single-step launches in the steady state of the benchmark workload (solver_residual_threshold %g, warm start %d)' % (DTYPE, n, steps, RESID, WARM))
print('sweeps per robot-step: mean %.2f, median %d, p90 %d, p99 %d; at the cap of %d: %.2f %%' % (sweeps.mean(), np.median(sweeps), np.percentile(sweeps, 90), np.percentile(sweeps, 99), cap, 100.0 * (sweeps >= cap).mean()))
edges = [0, 1, 2, 3, 5, 8, 12, 16, 24, 32, 40, 49, cap, cap + 1]
h = np.histogram(sweeps, bins=edges)[0]
print('  histogram: ' + '  '.join('[%d,%d): %.1f%%' % (edges[i], edges[i + 1], 100.0 * h[i] / len(sweeps)) for i in range(len(h))))
print('row updates per robot-step: mean %.1f; share of all row updates done by robot-steps at the cap: %.1f %%; share of all sweeps: %.1f %%' % (
  rows.mean(), 100.0 * rows[sweeps >= cap].sum() / max(rows.sum(), 1), 100.0 * sweeps[sweeps >= cap].sum() / max(sweeps.sum(), 1)))
m = sweeps >= cap
if m.any():
  per = rows[m] / sweeps[m]
  print('robot-steps at the cap: rows moved per sweep: mean %.2f, quartiles %s; touching spheres: %s' % (
    per.mean(), np.round(np.percentile(per, [25, 50, 75, 100]), 2).tolist(), {int(c): '%.0f%%' % (100.0 * (spheres[m] == c).mean()) for c in np.unique(spheres[m])}))
  hh = np.histogram(per, bins=[0, 1, 2, 3, 4, 6, 8, 12, 64])[0]
  print('  rows per sweep histogram [0,1) [1,2) [2,3) [3,4) [4,6) [6,8) [8,12) [12,..): %s %%' % np.round(100.0 * hh / m.sum(), 1).tolist())
for c in range(0, 13):
  mm = spheres == c
  if mm.any():
    print('  touching spheres %2d: %5.1f %% of robot-steps, mean sweeps %.1f, at the cap %.1f %%' % (c, 100.0 * mm.mean(), sweeps[mm].mean(), 100.0 * (sweeps[mm] >= cap).mean()))
env._close()
