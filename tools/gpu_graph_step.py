"""MEASUREMENT: Solo8VanillaEnv.step() captured into a HIP graph (torch.cuda.CUDAGraph) - the closed loop as an RL
library would capture it together with its policy - against the eager call: results bit for bit, and the time per step
of (a) eager step(), (b) a replayed graph of ONE step, (c) a replayed graph of 20 steps reading their actions from a
static [20, N, 12] buffer.  The engine enqueues everything on the caller's stream and never synchronises inside
step(), so the capture needs nothing special."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from gym_solo_amd import abi
from bench import build_env, desynchronise_episodes

N = 4096
for dtype in ('float32', 'float64'):
  tdt = torch.float32 if dtype == 'float32' else torch.float64
  envs = [build_env(N, 0, dtype) for _ in range(2)]   # [0]: eager, [1]: graph replay
  g = torch.Generator(device='cuda').manual_seed(77)
  for e in envs:
    desynchronise_episodes(e.engine, torch.Generator(device='cuda').manual_seed(5))
  acts = (torch.rand(200, N, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * (2 * np.pi)
  static_one = torch.zeros(N, 12, device='cuda', dtype=tdt)
  static_20 = torch.zeros(20, N, 12, device='cuda', dtype=tdt)
  env = envs[1]
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(side):   # (torch's capture recipe: warm up on a side stream)
    snap = env.engine.state.clone(); cnt = env.engine.term_count.clone()
    for _ in range(3):
      env.step(static_one)
    env.engine.state.copy_(snap); env.engine.term_count.copy_(cnt)
  torch.cuda.current_stream().wait_stream(side)
  g1 = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g1):
    o1, r1, d1, _ = env.step(static_one)
  g20 = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g20):
    for i in range(20):
      env.step(static_20[i])
  env.engine.state.copy_(snap); env.engine.term_count.copy_(cnt)   # (the captures above did not run anything)
  # ---- equality: 40 steps eager vs 20 one-step replays + one 20-step replay
  worst = 0.0
  for k in range(20):
    oe, re_, de, _ = envs[0].step(acts[k])
    static_one.copy_(acts[k]); g1.replay()
    assert torch.equal(oe, o1) and torch.equal(re_, r1) and torch.equal(de, d1), (dtype, k)
  for k in range(20, 40):
    envs[0].step(acts[k])
  static_20.copy_(acts[20:40]); g20.replay()
  torch.cuda.synchronize()
  assert torch.equal(envs[0].engine.state, env.engine.state) and torch.equal(envs[0].engine.obs, env.engine.obs), dtype
  print('%s: 20 one-step graph replays + one 20-step graph replay == 40 eager step() calls, bit for bit (state, obs, reward, done)' % dtype, flush=True)

  def timeit(fn, steps):
    ts = []
    for _ in range(7):
      torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
      ts.append((time.perf_counter() - t0) / steps)
    return statistics.median(ts)

  def eager():
    for k in range(100):
      env.step(acts[k])
  def replay_one():
    for k in range(100):
      g1.replay()
  def replay_20():
    for k in range(5):
      g20.replay()
  for name, fn in (('eager step()', eager), ('graph of 1 step, replayed', replay_one), ('graph of 20 steps, replayed', replay_20), ('eager step() again', eager)):
    t = timeit(fn, 100)
    print('   %-30s %6.1f us per step = %.3g env-steps/s' % (name, t * 1e6, N / t), flush=True)
  for e in envs:
    e._close()
