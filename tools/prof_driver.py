"""Runs the bench workload without timing code, for rocprofv3 passes (not a test).
env: STEPS (steps per rollout, default 2000), REPEATS (rollouts, default 1), DTYPE, N; SPL / STREAMS / MIGRATE override the
launch geometry, which is otherwise THE ENGINE'S (SoloConfig's -1 defaults, as bench.py runs: Engine.plan).
The driver's geometry: STEPS=20 REPEATS=40 (one 4096 x 20 launch per rollout)."""
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import build_env, desynchronise_episodes
from gym_solo_amd import abi
dtype = os.environ.get('DTYPE', 'float64'); n = int(os.environ.get('N', '4096'))
steps = int(os.environ.get('STEPS', '2000'))
env = build_env(n, 0, dtype, steps_per_launch=int(os.environ.get('SPL', '-1')), rollout_streams=int(os.environ.get('STREAMS', '-1')),
                migrate_steps=int(os.environ.get('MIGRATE', '-1')))
eng = env.engine
plan = eng.plan(steps)
tdt = torch.float32 if dtype == 'float32' else torch.float64
g = torch.Generator(device='cuda').manual_seed(1234)
desynchronise_episodes(eng, g, chunk=steps if plan['launches'] == 1 else 2 * plan['steps_per_launch'])  # as bench.py does: terminations / auto-resets inside every window; launches of the measured geometry
acts = (torch.rand(steps, n, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * (2 * np.pi)
out = eng.rollout_buffers(acts.shape[0])
for _ in range(int(os.environ.get('REPEATS', '1'))):   # (short geometries: several launches to average over)
  eng.rollout(acts, abi.STEP_ALL, out=out)
torch.cuda.synchronize()
meta = {'robots_per_launch': n // plan['slices'], 'steps_per_launch': plan['steps_per_launch'], 'steps': steps, 'dtype': dtype,
        'launch_chains': plan['slices'], 'migrate_steps': plan['migrate_steps'], 'plan': plan}
if len(sys.argv) > 1:
  json.dump(meta, open(sys.argv[1], 'w'))
print('done', eng.kernel_name, meta)
