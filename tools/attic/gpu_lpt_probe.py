"""MEASUREMENT: what is a longest-first launch order worth to the f64 kernel at three waves per SIMD?  4096 robots on
3072 wave slots: the last quarter of the workgroups starts late, and a slow robot among them ends the launch late.
Host-side orders (set_order before every launch, outside the timed region) as the upper bound for a device-side one:
identity | full sort by the previous launch's sweeps | two classes (sweeps per step above a threshold first).
usage: gpu_lpt_probe.py [float64] [K]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gym_solo_amd import abi
from bench import build_env, desynchronise_episodes
dtype = sys.argv[1] if len(sys.argv) > 1 else 'float64'
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 4096
tdt = torch.float32 if dtype == 'float32' else torch.float64
env = build_env(n, 0, dtype, steps_per_launch=k, rollout_streams=1, migrate_steps=int(os.environ.get('MIGRATE', '0')))
print('migrate_steps', os.environ.get('MIGRATE', '0'))
eng = env.engine
g = torch.Generator(device='cuda').manual_seed(1234)
desynchronise_episodes(eng, g)
pool = (torch.rand(40 * k, n, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * (2 * np.pi)
out = eng.rollout_buffers(k)
eng.rollout(pool[:k], abi.STEP_ALL, out=out)
def two_class(thr):
  c = eng.cost
  exp = (c > thr * k)
  idx = torch.arange(n, device='cuda', dtype=torch.int32)
  return torch.cat([idx[exp], idx[~exp]]).contiguous(), float(exp.float().mean())
for mode in ('identity', 'sort', 'two-class 4', 'two-class 6', 'two-class 8', 'two-class 12', 'identity', 'sort'):
  eng.set_order(None)
  ts, frac = [], 0.0
  for rep in range(30):
    a = pool[(rep % 40) * k:(rep % 40 + 1) * k]
    if mode == 'sort':
      eng.balance()
    elif mode.startswith('two-class'):
      o, frac = two_class(int(mode.split()[1]))
      eng.set_order(o)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.rollout(a, abi.STEP_ALL, out=out)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
  print('%s N=%d K=%d %-14s %.4g env-steps/s (median of 30; expensive share %.2f)' % (dtype, n, k, mode, n * k / np.median(ts), frac), flush=True)
c = eng.cost.cpu().numpy() / k
print('sweeps per step, last launch: mean %.2f, quantiles 50/75/90/99/max: %s' % (c.mean(), np.quantile(c, [.5, .75, .9, .99, 1.0]).round(2).tolist()))
env._close()
