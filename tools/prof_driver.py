"""Runs the bench workload without timing code, for rocprofv3 passes (not a test).
env: SPL (steps per launch, default 250), STREAMS (default 2), STEPS (default 2000), DTYPE, N, REPEATS.
The driver's geometry: SPL=20 STREAMS=1 STEPS=20 REPEATS=40 (one 4096 x 20 launch per rollout)."""
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import build_env, desynchronise_episodes
from gym_solo_amd import abi
dtype = os.environ.get('DTYPE', 'float32'); n = int(os.environ.get('N', '4096'))
spl, streams, steps = int(os.environ.get('SPL', '250')), int(os.environ.get('STREAMS', '2')), int(os.environ.get('STEPS', '2000'))
# robot migration as bench.py chooses it (-1 = its rule: half the launch for a single-launch f64 rollout, else off)
migrate = int(os.environ.get('MIGRATE', '-1'))
if migrate < 0:
  migrate = 0
  if dtype == 'float64' and spl >= 8:
    if steps <= spl:
      migrate = (min(spl, steps) + 1) // 2
    elif spl >= 50:
      migrate, streams = 25, 1
env = build_env(n, 0, dtype, steps_per_launch=spl, rollout_streams=streams, migrate_steps=migrate)
eng = env.engine
tdt = torch.float32 if dtype == 'float32' else torch.float64
g = torch.Generator(device='cuda').manual_seed(1234)
desynchronise_episodes(eng, g)  # as bench.py does: terminations / auto-resets inside every window
acts = (torch.rand(steps, n, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * (2 * np.pi)
out = eng.rollout_buffers(acts.shape[0])
for _ in range(int(os.environ.get('REPEATS', '1'))):   # (short geometries: several launches to average over)
  eng.rollout(acts, abi.STEP_ALL, out=out)
torch.cuda.synchronize()
meta = {'robots_per_launch': n // streams if streams > 1 else n, 'steps_per_launch': min(spl, steps), 'steps': steps, 'dtype': dtype,
        'launch_chains': streams if steps > spl else 1, 'migrate_steps': migrate}
if len(sys.argv) > 1:
  json.dump(meta, open(sys.argv[1], 'w'))
print('done', eng.kernel_name, meta)
