"""GPU probe: wall time of a 1000-step rollout vs host enqueue time vs kernel time (not a test)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from gym_solo_amd import abi
from bench import build_env
n = 4096
env = build_env(n, 0, 'float32'); eng = env.engine
g = torch.Generator(device='cuda').manual_seed(1234)
acts = (torch.rand(1000, n, 12, device='cuda', dtype=torch.float32, generator=g) * 2 - 1) * (2 * np.pi)
eng.rollout(acts[:100], abi.STEP_ALL); torch.cuda.synchronize()
for rep in range(3):
  t0 = time.perf_counter(); eng.rollout(acts, abi.STEP_ALL); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
  print(f'rollout 1000: host enqueue {1e3*(t1-t0):.2f} ms  total {1e3*(t2-t0):.2f} ms -> {(t2-t0)*1e3:.1f} us/step', flush=True)
ms = eng.time_step(acts[:500], abi.STEP_ALL)
print(f'time_step same action x500: {ms*1e3:.1f} us/launch')
t0 = time.perf_counter()
for i in range(300): eng.step(acts[i], abi.STEP_ALL)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'python engine.step x300: host {1e6*(t1-t0)/300:.1f} us/call  total {1e6*(t2-t0)/300:.1f} us/step')
t0 = time.perf_counter()
for i in range(300): env.step(acts[i])
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'python env.step x300: host {1e6*(t1-t0)/300:.1f} us/call  total {1e6*(t2-t0)/300:.1f} us/step')
env._close()
for streams in (2, 4):
  os.environ['SOLO_ROLLOUT_STREAMS'] = str(streams)
  env = build_env(n, 0, 'float32'); eng = env.engine
  eng.rollout(acts[:100], abi.STEP_ALL); torch.cuda.synchronize()
  for rep in range(2):
    t0 = time.perf_counter(); eng.rollout(acts, abi.STEP_ALL); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'SOLO_ROLLOUT_STREAMS={streams}: rollout 1000 total {1e3*(t2-t0):.2f} ms -> {(t2-t0)*1e3:.1f} us/step  {n*1000/(t2-t0):.3e} env-steps/s', flush=True)
  env._close()
