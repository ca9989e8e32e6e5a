"""MEASUREMENT: how well does a robot's Gauss-Seidel cost in one K-step launch predict its cost in the NEXT one?  (The
launch order / placement can only use history: view.cost = sweeps of the last launch.)  Benchmark workload, steady state,
consecutive K-step launches; prints the rank correlation and how many of the next launch's costliest robots were among
the costliest of the previous one.
  python tools/gpu_cost_persistence.py [float64] [20] [4096]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from gym_solo_amd import abi

dtype = sys.argv[1] if len(sys.argv) > 1 else 'float64'
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
tdt = torch.float32 if dtype == 'float32' else torch.float64
env = bench.build_env(n, 0, dtype, steps_per_launch=k, rollout_streams=1, migrate_steps=0)
eng = env.engine
gen = torch.Generator(device='cuda').manual_seed(1234)
bench.desynchronise_episodes(eng, gen)
pool = lambda steps: (torch.rand(steps, n, abi.NUM_JOINTS, device='cuda', dtype=tdt, generator=gen) * 2 - 1) * 6.283185307179586
costs = []
for rep in range(12):
  eng.rollout(pool(k), abi.STEP_ALL)
  torch.cuda.synchronize()
  costs.append(eng.cost.cpu().numpy().astype(np.int64).copy())
costs = np.array(costs)
print('%s N = %d K = %d: sweeps per robot per launch: mean %.0f, p50 %.0f, p90 %.0f, p99 %.0f, max %d (cap %d)' % (
  dtype, n, k, costs.mean(), np.percentile(costs, 50), np.percentile(costs, 90), np.percentile(costs, 99), costs.max(), 50 * k))
def ranks(x):
  r = np.empty_like(x); r[np.argsort(x, kind='stable')] = np.arange(len(x)); return r
rho = [np.corrcoef(ranks(costs[t]), ranks(costs[t + 1]))[0, 1] for t in range(len(costs) - 1)]
print('rank correlation of consecutive launches: mean %.2f (min %.2f, max %.2f)' % (np.mean(rho), np.min(rho), np.max(rho)))
for top in (16, 64, 128, 256):
  for pool_size in (64, 128, 256, 512, 1024):
    if pool_size < top:
      continue
    hit = []
    for t in range(len(costs) - 1):
      nxt = set(np.argsort(-costs[t + 1], kind='stable')[:top].tolist())
      prev = set(np.argsort(-costs[t], kind='stable')[:pool_size].tolist())
      hit.append(len(nxt & prev) / top)
    print('  of the %3d costliest robots of a launch, %.0f %% were among the %4d costliest of the previous launch' % (top, 100 * np.mean(hit), pool_size))
# how costly is the costliest robot that the previous launch did NOT flag?
for pool_size in (64, 128, 256, 512):
  worst = []
  for t in range(len(costs) - 1):
    prev = np.argsort(-costs[t], kind='stable')[:pool_size]
    m = np.ones(n, bool); m[prev] = False
    worst.append(costs[t + 1][m].max() / costs[t + 1].max())
  print('  the costliest robot NOT among the previous launch\'s %3d costliest runs %.2f of the launch\'s maximum (mean; min %.2f, max %.2f)' % (pool_size, np.mean(worst), np.min(worst), np.max(worst)))
env._close()


def in_launch_persistence(first=2, rest=18, reps=8):
  """Does a robot's cost over the FIRST `first` steps of a launch predict its cost over the remaining `rest`?  (Launches of
  `first` and `rest` steps alternate; view.cost is the sweeps of the last launch.)"""
  env = bench.build_env(n, 0, dtype, steps_per_launch=max(first, rest), rollout_streams=1, migrate_steps=0)
  eng = env.engine
  gen = torch.Generator(device='cuda').manual_seed(1234)
  bench.desynchronise_episodes(eng, gen)
  pool = lambda steps: (torch.rand(steps, n, abi.NUM_JOINTS, device='cuda', dtype=tdt, generator=gen) * 2 - 1) * 6.283185307179586
  rho, hits = [], {}
  for rep in range(reps):
    eng.rollout(pool(first), abi.STEP_ALL); torch.cuda.synchronize()
    a = eng.cost.cpu().numpy().astype(np.int64).copy()
    eng.rollout(pool(rest), abi.STEP_ALL); torch.cuda.synchronize()
    b = eng.cost.cpu().numpy().astype(np.int64).copy()
    rho.append(np.corrcoef(ranks(a), ranks(b))[0, 1])
    for top, pool_size in ((64, 256), (64, 512), (64, 1024), (256, 1024)):
      nxt = set(np.argsort(-b, kind='stable')[:top].tolist()); prev = set(np.argsort(-a, kind='stable')[:pool_size].tolist())
      hits.setdefault((top, pool_size), []).append(len(nxt & prev) / top)
  print('IN-LAUNCH: sweeps of the first %d steps vs the following %d: rank correlation %.2f (min %.2f, max %.2f)' % (first, rest, np.mean(rho), np.min(rho), np.max(rho)))
  for (top, pool_size), h in hits.items():
    print('  of the %3d costliest robots over the following %d steps, %.0f %% were among the %4d costliest of the first %d' % (top, rest, 100 * np.mean(h), pool_size, first))
  env._close()


in_launch_persistence(2, 18)
in_launch_persistence(4, 16)
in_launch_persistence(10, 10)
