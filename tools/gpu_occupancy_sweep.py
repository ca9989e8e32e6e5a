"""MEASUREMENT: the step kernel's launch duration and throughput at the driver's geometry (ONE K-step launch, steady
state of the benchmark workload) over a range of batch sizes - how the wave slots of the chip fill (1024 SIMDs x W
waves: W = 4 in f32, 3 in f64) and what a batch that is not a multiple of them costs.
  SOLO_HIP_LIB=... python tools/gpu_occupancy_sweep.py float64 20 2048 3072 4096 6144 8192"""
import sys, os, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gym_solo_amd import abi

dtype = sys.argv[1] if len(sys.argv) > 1 else 'float64'
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sizes = [int(x) for x in sys.argv[3:]] or [2048, 3072, 4096, 6144, 8192]
tdt = torch.float32 if dtype == 'float32' else torch.float64
print('library: %s  migrate_steps: %s' % (os.environ.get('SOLO_HIP_LIB', 'libsolo_hip.so'), os.environ.get('MIGRATE', '0')), flush=True)
for n in sizes:
  env = bench.build_env(n, 0, dtype, steps_per_launch=k, rollout_streams=1, migrate_steps=int(os.environ.get('MIGRATE', '0')))
  eng = env.engine
  gen = torch.Generator(device='cuda').manual_seed(1234)
  bench.desynchronise_episodes(eng, gen)
  def pool(steps):
    return (torch.rand(steps, n, abi.NUM_JOINTS, device='cuda', dtype=tdt, generator=gen) * 2 - 1) * 6.283185307179586
  eng.rollout(pool(k), abi.STEP_ALL, out=eng.rollout_buffers(k))
  ms = statistics.median(eng.time_step(pool(k), abi.STEP_ALL) for _ in range(9))
  print('%s  N = %5d  K = %d: kernel %.4f ms  -> %.4g env-steps/s  (%.2f us per robot-step-slot at %d waves)' % (
    dtype, n, k, ms, n * k / (ms * 1e-3), ms * 1e3 / k, n), flush=True)
  env._close()
