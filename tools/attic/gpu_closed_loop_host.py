"""MEASUREMENT: host time per closed-loop step (python -> ctypes -> one kernel launch) against the
device time per step, N = 4096, f32, one solo_engine_step launch per env step."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gym_solo_amd import abi
from bench import build_env
n, k = 4096, 500
env = build_env(n, 0, 'float32', steps_per_launch=1, rollout_streams=1)
eng = env.engine
g = torch.Generator(device='cuda').manual_seed(1234)
acts = (torch.rand(k, n, 12, device='cuda', generator=g) * 2 - 1) * (2 * np.pi)
from bench import desynchronise_episodes
desynchronise_episodes(eng, g)     # steady state first: the cost of a step depends on the episode phase
def raw():
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for i in range(k):
    eng.step(acts[i], abi.STEP_ALL)
  t1 = time.perf_counter()            # every launch is enqueued
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  print('raw solo_engine_step, %d steps: host enqueue %.1f us/step, device-bound total %.1f us/step -> %.4g env-steps/s' % (
    k, (t1 - t0) / k * 1e6, (t2 - t0) / k * 1e6, n * k / (t2 - t0)), flush=True)
def api():
  # Solo8VanillaEnv.step (the reference's API; zero-copy outputs)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for i in range(k):
    env.step(acts[i])
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  print('Solo8VanillaEnv.step,  %d steps: host enqueue %.1f us/step, device-bound total %.1f us/step -> %.4g env-steps/s' % (
    k, (t1 - t0) / k * 1e6, (t2 - t0) / k * 1e6, n * k / (t2 - t0)), flush=True)
# alternating, on the same steady-state workload (round 2 ran the two one after the other on a drifting one and
# read the drift as API overhead)
for rep in range(3):
  raw()
  api()
