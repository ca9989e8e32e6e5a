"""MEASUREMENT: what the host adds to a K = 20 timed region (f64, the driver's geometry): wall clock of rollout + device
synchronisation against the launch chain's HIP events, for three ways of waiting - torch.cuda.synchronize() alone, a
query spin (hipStreamQuery through torch's stream.query()) in front of it, and an event-query spin.
  python tools/gpu_sync_latency.py            (ROC_ACTIVE_WAIT_TIMEOUT=us in the environment changes the runtime's own wait)"""
import sys, os, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from gym_solo_amd import abi
k, n, dtype = 20, 4096, 'float64'
env = bench.build_env(n, 0, dtype, steps_per_launch=k, rollout_streams=1, migrate_steps=10)
eng = env.engine
gen = torch.Generator(device='cuda').manual_seed(1234)
bench.desynchronise_episodes(eng, gen)
out = eng.rollout_buffers(k)
stream = torch.cuda.current_stream()
print('ROC_ACTIVE_WAIT_TIMEOUT =', os.environ.get('ROC_ACTIVE_WAIT_TIMEOUT'))
for mode in ('synchronize', 'query spin', 'event spin', 'synchronize', 'query spin', 'event spin'):
  ev, wall = [], []
  for rep in range(30):
    acts = (torch.rand(k, n, abi.NUM_JOINTS, device='cuda', dtype=torch.float64, generator=gen) * 2 - 1) * 6.283185307179586
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e0.record()
    eng.rollout(acts, abi.STEP_ALL, out=out)
    e1.record()
    if mode == 'query spin':
      while not stream.query():
        pass
    elif mode == 'event spin':
      while not e1.query():
        pass
    torch.cuda.synchronize(); wall.append((time.perf_counter() - t0) * 1e3)
    ev.append(e0.elapsed_time(e1))
  print('%-12s events %.3f ms, wall %.3f ms (min %.3f): the host adds %.0f us' % (mode, statistics.median(ev), statistics.median(wall), min(wall), 1e3 * (statistics.median(wall) - statistics.median(ev))), flush=True)
env._close()
