"""MEASUREMENT: throughput of the BASELINE configurations on one MI355X - configs[1] (4096 envs, flat), configs[3] (8192
envs/GPU, per-env friction + base-mass randomisation), configs[4] (4096 envs/GPU on the incline / stairs heightfields) -
on the rollout path of bench.py (every step's outputs recorded), in the steady state of the workload (episode phases
spread, 1000 untimed steps), at the driver's geometry (K = 20) and the default one (K = 1000), with the launch geometry
the ENGINE chooses (Engine.plan) - and, for the 8192-robot configuration, against the alternatives.
  python tools/gpu_configs.py [--dtype float64]"""
import argparse, sys, os, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import helpers
import bench
from gym_solo_amd import abi
from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig, Solo8VanillaEnv
from gym_solo_amd.workloads import register_benchmark_workload

ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='float64')
args = ap.parse_args()
tdt = torch.float32 if args.dtype == 'float32' else torch.float64


def run(name, n, terrain=None, randomise=False, **knobs):
  cfg = Solo8VanillaConfig()
  cfg.num_envs, cfg.dtype, cfg.auto_reset = n, args.dtype, True
  for k_, v in knobs.items():
    setattr(cfg, k_, v)
  if terrain is not None:
    cfg.terrain = terrain
  env = Solo8VanillaEnv(config=cfg, copy_outputs=False)
  register_benchmark_workload(env, max_steps=1000)
  env._ensure_program()
  eng = env.engine
  g = torch.Generator(device='cuda').manual_seed(4321)
  if randomise:
    eng.set_params(abi.PARAM_FRICTION, torch.rand(n, device='cuda', dtype=tdt, generator=g) * 0.7 + 0.3)
    eng.set_params(abi.PARAM_BASE_MASS_SCALE, torch.rand(n, device='cuda', dtype=tdt, generator=g) * 0.4 + 0.8)
    eng.settle()
  bench.desynchronise_episodes(eng, g)
  res = []
  for k, reps in ((20, 30), (1000, 4)):
    out = eng.rollout_buffers(k)
    pool = lambda: (torch.rand(k, n, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * (2 * np.pi)
    eng.rollout(pool(), abi.STEP_ALL, out=out)
    ts = []
    for rep in range(reps):
      a = pool()
      torch.cuda.synchronize(); t0 = time.perf_counter()
      eng.rollout(a, abi.STEP_ALL, out=out)
      torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    p = eng.plan(k)
    res.append('K = %d: %.4g env-steps/s (%d x %d steps, %d slice(s), migrate %d)' % (k, n * k / statistics.median(ts), p['launches'], p['steps_per_launch'], p['slices'], p['migrate_steps']))
  st = eng.stats.cpu().numpy()
  print('%-78s %s; diverged %d, waves that gave up %d' % (name, '; '.join(res), st[5], st[6]), flush=True)
  env._close()


print('%s, one MI355X, steady state, every step recorded; median of 30 (K = 20) / 4 (K = 1000) repeats' % args.dtype)
run('configs[1]  4096 envs, flat plane', 4096)
run('configs[3]  8192 envs, friction U(.3,1) + base mass U(.8,1.2)', 8192, randomise=True)
run('configs[3]  ... no migration, one chain', 8192, randomise=True, migrate_steps=0, rollout_streams=1)
run('configs[3]  ... no migration, two slices', 8192, randomise=True, migrate_steps=0, rollout_streams=2)
run('configs[3]  ... migration in chunks of 5 / one chain', 8192, randomise=True, migrate_steps=5, rollout_streams=1)
run('configs[3]  ... migration in chunks of 25 / one chain', 8192, randomise=True, migrate_steps=25, rollout_streams=1)
run('configs[4]  4096 envs, 10 degree incline heightfield', 4096, terrain=helpers.incline_terrain())
run('configs[4]  4096 envs, stairs 0.03 m x 0.30 m heightfield', 4096, terrain=helpers.stairs_terrain())
