"""MEASUREMENT: what a K = 20 timed repeat of bench.py is made of besides the step kernel - host launch, the
statistics reduction, and the host's wake-up from the device synchronisation that the bench contract puts on both
sides of every repeat.  usage: gpu_k20_segments.py [float32|float64]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gym_solo_amd import abi
from bench import build_env, desynchronise_episodes
dtype = (sys.argv[1:] or ['float32'])[0]
tdt = torch.float32 if dtype == 'float32' else torch.float64
n, k = 4096, 20
env = build_env(n, 0, dtype, steps_per_launch=k, rollout_streams=1)
eng = env.engine
g = torch.Generator(device='cuda').manual_seed(1234)
desynchronise_episodes(eng, g)
pool = (torch.rand(40 * k, n, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * (2 * np.pi)
out = eng.rollout_buffers(k)
def med(f, reps=40):
  ts = []
  for r in range(reps):
    ts.append(f(r))
  return 1e6 * float(np.median(ts))
def sync(): torch.cuda.synchronize()
def t_rollout_sync(r):
  a = pool[(r % 40) * k:(r % 40 + 1) * k]; sync(); t0 = time.perf_counter(); eng.rollout(a, abi.STEP_ALL, out=out); sync(); return time.perf_counter() - t0
def t_rollout_stats_sync(r):
  a = pool[(r % 40) * k:(r % 40 + 1) * k]; sync(); t0 = time.perf_counter(); eng.rollout(a, abi.STEP_ALL, out=out); s = eng.stats_shards.sum(dim=0); sync(); return time.perf_counter() - t0
def t_events(r):
  a = pool[(r % 40) * k:(r % 40 + 1) * k]; e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); sync()
  e0.record(); eng.rollout(a, abi.STEP_ALL, out=out); e1.record(); sync(); return e0.elapsed_time(e1) * 1e-3
def t_empty_sync(r):
  sync(); t0 = time.perf_counter(); sync(); return time.perf_counter() - t0
x = torch.zeros(64, device='cuda')
def t_tiny_kernel_sync(r):
  sync(); t0 = time.perf_counter(); x.add_(1.0); sync(); return time.perf_counter() - t0
def t_launch_only(r):
  a = pool[(r % 40) * k:(r % 40 + 1) * k]; sync(); t0 = time.perf_counter(); eng.rollout(a, abi.STEP_ALL, out=out); t = time.perf_counter() - t0; sync(); return t
for f in (t_rollout_sync, t_rollout_sync, t_rollout_stats_sync, t_events, t_empty_sync, t_tiny_kernel_sync, t_launch_only):
  print('%-24s %8.1f us (median of 40)' % (f.__name__, med(f)), flush=True)
print('step kernel alone (HIP events inside the engine): %.1f us' % (1e3 * np.median([eng.time_step(pool[:k], abi.STEP_ALL) for _ in range(5)])))
