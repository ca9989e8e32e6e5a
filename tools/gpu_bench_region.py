"""MEASUREMENT: what bench.py's K = 20 timed region is made of (f64, the engine's geometry): the launch's HIP-event duration
against the wall clock between the two device synchronisations, for the bench's own repeat (fresh actions + a statistics
read in front of the first synchronisation), a tight loop, and a repeat that lets the GPU idle for 2 ms first.
  python tools/gpu_bench_region.py"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gym_solo_amd import abi
n, k = 4096, 20
env = bench.build_env(n, 0, 'float64')
eng = env.engine
gen = torch.Generator(device='cuda').manual_seed(1234)
bench.desynchronise_episodes(eng, gen, chunk=k)
out = eng.rollout_buffers(k)
pool = lambda: (torch.rand(k, n, abi.NUM_JOINTS, device='cuda', dtype=torch.float64, generator=gen) * 2 - 1) * 6.283185307179586
def repeat(mode):
  a = pool()
  if mode != 'tight':
    before = eng.stats_shards.sum(dim=0)
  torch.cuda.synchronize()
  if mode == 'idle 2 ms':
    time.sleep(0.002)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  e0.record()
  eng.rollout(a, abi.STEP_ALL, out=out)
  e1.record()
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  return (t2 - t0) * 1e3, (t1 - t0) * 1e3, e0.elapsed_time(e1)
for mode in ('bench', 'tight', 'idle 2 ms', 'bench', 'tight', 'idle 2 ms'):
  r = [repeat(mode) for _ in range(40)]
  print('%-10s wall %.3f ms (min %.3f), of it the host\'s enqueue %.3f ms; HIP events around the launch %.3f ms -> %.4g env-steps/s' % (
    mode, statistics.median(x[0] for x in r), min(x[0] for x in r), statistics.median(x[1] for x in r), statistics.median(x[2] for x in r), n * k / statistics.median(x[0] for x in r) * 1e3), flush=True)
print('time_rollout: %.3f ms' % statistics.median(eng.time_rollout(pool(), abi.STEP_ALL) for _ in range(9)))
env._close()
