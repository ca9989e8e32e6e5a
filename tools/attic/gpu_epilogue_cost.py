"""MEASUREMENT: what the output epilogue (observations / rewards / done / returns of a launch, evaluated by the robot's
own wave: solo_step_kernel.h) costs at the driver's geometry - the same K-step launch with the outputs switched off one
by one.  Kernel time by HIP events, steady state of the benchmark workload, median of 9.
  MIGRATE=10 python tools/gpu_epilogue_cost.py float64 20 4096"""
import sys, os, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from gym_solo_amd import abi

dtype = sys.argv[1] if len(sys.argv) > 1 else 'float64'
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sizes = [int(x) for x in sys.argv[3:]] or [4096]
tdt = torch.float32 if dtype == 'float32' else torch.float64
mig = int(os.environ.get('MIGRATE', '0'))
P, O, R, D = abi.STEP_PHYSICS, abi.STEP_OBS, abi.STEP_REWARD, abi.STEP_DONE
for n in sizes:
  env = bench.build_env(n, 0, dtype, steps_per_launch=k, rollout_streams=1, migrate_steps=mig)
  eng = env.engine
  gen = torch.Generator(device='cuda').manual_seed(1234)
  bench.desynchronise_episodes(eng, gen)
  def pool(steps):
    return (torch.rand(steps, n, abi.NUM_JOINTS, device='cuda', dtype=tdt, generator=gen) * 2 - 1) * 6.283185307179586
  eng.rollout(pool(k), abi.STEP_ALL, out=eng.rollout_buffers(k))
  for name, flags in (('all outputs', P | O | R | D), ('physics + done + reward', P | R | D), ('physics + done + observations', P | O | D),
                      ('physics + done', P | D), ('physics only', P), ('all outputs (again)', P | O | R | D)):
    ms = statistics.median(eng.time_step(pool(k), flags) for _ in range(9))
    print('%s  N = %5d  K = %d  migrate %2d  %-30s kernel %.4f ms' % (dtype, n, k, mig, name, ms), flush=True)
  env._close()
