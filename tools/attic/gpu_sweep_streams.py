"""GPU probe: rollout throughput vs steps_per_launch x SOLO_ROLLOUT_STREAMS (not a test)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from gym_solo_amd import abi
from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig, Solo8VanillaEnv
from gym_solo_amd.workloads import register_benchmark_workload
n = 4096
g = torch.Generator(device='cuda').manual_seed(1234)
acts = (torch.rand(1000, n, 12, device='cuda', dtype=torch.float32, generator=g) * 2 - 1) * (2 * np.pi)
for streams in (1, 2, 3, 4, 8):
  for spl in (1, 25, 100, 250, 1000):
    cfg = Solo8VanillaConfig()
    cfg.num_envs, cfg.dtype, cfg.auto_reset, cfg.steps_per_launch, cfg.rollout_streams = n, 'float32', True, spl, streams
    env = Solo8VanillaEnv(config=cfg, copy_outputs=False)
    register_benchmark_workload(env, max_steps=1000); env._ensure_program()
    eng = env.engine
    eng.rollout(acts[:100], abi.STEP_ALL); torch.cuda.synchronize()
    best = 1e9
    for rep in range(2):
      t0 = time.perf_counter(); eng.rollout(acts, abi.STEP_ALL); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f'streams={streams} steps_per_launch={spl}: {best*1e3:.1f} us/step  {n*1000/best:.3e} env-steps/s', flush=True)
    env._close()
