"""MEASUREMENT: cost-balanced launch order (Engine.balance: costliest robots first) vs the identity,
for batches at and above the chip's 4096 resident waves.  usage: gpu_balance.py [N ...]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gym_solo_amd import abi
from bench import build_env
spl, k = 250, 1500
for n in [int(a) for a in sys.argv[1:]] or (4096, 8192, 16384):
  for streams in (1, 2):
    env = build_env(n, 0, 'float32', steps_per_launch=spl, rollout_streams=streams)
    eng = env.engine
    g = torch.Generator(device='cuda').manual_seed(1234)
    acts = (torch.rand(k, n, 12, device='cuda', generator=g) * 2 - 1) * (2 * np.pi)
    eng.rollout(acts[:500], abi.STEP_ALL)
    res = {}
    for mode in ('identity', 'balanced', 'identity', 'balanced'):
      ts = []
      for rep in range(3):
        if mode == 'balanced':
          eng.balance()          # from the previous launch's sweeps (inside the timed region: it is part of the policy)
        else:
          eng.set_order(None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.rollout(acts, abi.STEP_ALL)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
      res.setdefault(mode, []).append(n * k / np.median(ts))
    print('N=%d streams=%d: identity %s  balanced (re-sorted before each %d-step rollout) %s env-steps/s' % (
      n, streams, ['%.4g' % x for x in res['identity']], k, ['%.4g' % x for x in res['balanced']]), flush=True)
    env._close()
