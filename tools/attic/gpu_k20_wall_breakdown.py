"""MEASUREMENT: what a K = 20 timed repeat of bench.py is made of in f64 - the launch chain by HIP events on the launch
stream (queue init + step kernel), with and without recording every step's outputs, against the wall clock of
rollout + device synchronisation."""
import sys, os, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from gym_solo_amd import abi
dtype = sys.argv[1] if len(sys.argv) > 1 else 'float64'
k, n = 20, 4096
tdt = torch.float32 if dtype == 'float32' else torch.float64
for m in (10, 0):
  env = bench.build_env(n, 0, dtype, steps_per_launch=k, rollout_streams=1, migrate_steps=m if dtype == 'float64' else 0)
  eng = env.engine
  gen = torch.Generator(device='cuda').manual_seed(1234)
  bench.desynchronise_episodes(eng, gen)
  out = eng.rollout_buffers(k)
  res = {}
  for mode in ('record', 'no record', 'record', 'no record'):
    ev, wall = [], []
    for rep in range(15):
      acts = (torch.rand(k, n, abi.NUM_JOINTS, device='cuda', dtype=tdt, generator=gen) * 2 - 1) * 6.283185307179586
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      torch.cuda.synchronize(); t0 = time.perf_counter()
      e0.record()
      eng.rollout(acts, abi.STEP_ALL, out=out if mode == 'record' else None)
      e1.record()
      torch.cuda.synchronize(); wall.append((time.perf_counter() - t0) * 1e3)
      ev.append(e0.elapsed_time(e1))
    res.setdefault(mode, []).append((statistics.median(ev), statistics.median(wall)))
  print('%s migrate_steps %d: ' % (dtype, m) + '; '.join('%s: events %.3f ms, wall %.3f ms' % (mode, *v[-1]) for mode, v in res.items()), flush=True)
  env._close()
