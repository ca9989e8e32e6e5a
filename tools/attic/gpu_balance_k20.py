"""MEASUREMENT: cost-balanced launch order (Engine.balance: costliest robots first) vs the identity on the
DRIVER's geometry - one K = 20 launch - where the batch exceeds the resident wave slots: f64 at N = 4096
(two waves per SIMD: 2048 slots, every slot runs two robots one after the other) and f32 at N = 8192.
usage: gpu_balance_k20.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gym_solo_amd import abi
from bench import build_env, desynchronise_episodes
k = 20
for dtype, n in (('float64', 4096), ('float32', 4096), ('float32', 8192), ('float64', 2048)):
  tdt = torch.float32 if dtype == 'float32' else torch.float64
  env = build_env(n, 0, dtype, steps_per_launch=k, rollout_streams=1)
  eng = env.engine
  g = torch.Generator(device='cuda').manual_seed(1234)
  desynchronise_episodes(eng, g)
  pool = (torch.rand(40 * k, n, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * (2 * np.pi)
  out = eng.rollout_buffers(k)
  eng.rollout(pool[:k], abi.STEP_ALL, out=out)
  res = {}
  for mode in ('identity', 'balanced once', 'identity', 'balanced every launch', 'balanced once'):
    eng.set_order(None)
    if mode == 'balanced once':
      eng.balance()
    ts = []
    for rep in range(30):
      a = pool[(rep % 40) * k:(rep % 40 + 1) * k]
      if mode == 'balanced every launch':
        eng.balance()
      torch.cuda.synchronize(); t0 = time.perf_counter()
      eng.rollout(a, abi.STEP_ALL, out=out)
      torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    res.setdefault(mode, []).append(n * k / np.median(ts))
  print('%s N=%d K=%d: %s' % (dtype, n, k, {m: ['%.4g' % x for x in v] for m, v in res.items()}), flush=True)
  env._close()
