"""BASELINE configs[2] readiness on real hardware: the env batch sharded over TWO GPUs, one process per
GPU, `init_process_group('nccl')` (= RCCL over xGMI) and the design's ONE collective - the sum
all-reduce of the episodic-return statistics.  The logic of tests/test_distributed_gloo.py
on the HIP engine and RCCL: the reduced statistics must equal the union of the two shards computed by
one process.  Needs >= 2 visible GPUs; skipped on the build's 1-GPU boxes (the driver's 8-GPU node runs it).

The ranks are fresh child processes (their own HIP runtime each); `torch.cuda.device_count()` does not
initialise the GPU in this process."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_PER_RANK, STEPS, MAX_STEPS = 64, 24, 7

_SHARD = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import numpy as np, torch

def run_shard(rank, device):
  from gym_solo_amd import abi
  from gym_solo_amd.distributed import rank_seed
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig, Solo8VanillaEnv
  from gym_solo_amd.workloads import register_benchmark_workload
  cfg = Solo8VanillaConfig()
  cfg.dtype, cfg.num_envs, cfg.auto_reset, cfg.device, cfg.steps_per_launch = 'float64', %(n)d, True, device, 8
  env = Solo8VanillaEnv(config=cfg)
  register_benchmark_workload(env, max_steps=%(max_steps)d)
  env._ensure_program()
  g = torch.Generator(device='cuda:%%d' %% device).manual_seed(rank_seed(1234, rank))
  acts = (torch.rand(%(steps)d, %(n)d, 12, device='cuda:%%d' %% device, dtype=torch.float64, generator=g) * 2 - 1) * (2 * np.pi)
  env.engine.rollout(acts, abi.STEP_ALL)
  env.engine.synchronize()
  stats = env.engine.stats.clone()
  state = env.engine.state.cpu().numpy().copy()
  env._close()
  return stats, state
'''

_RANK = _SHARD + r'''
import torch.distributed as dist
rank, world, out = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), sys.argv[1]
torch.cuda.set_device(rank)
dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
from gym_solo_amd.distributed import all_reduce_stats
local, state = run_shard(rank, rank)
total = all_reduce_stats(local)
dist.barrier()
np.savez(out, local=local.cpu().numpy(), total=total.cpu().numpy(), state=state, backend=np.array(dist.get_backend()))
dist.destroy_process_group()
'''

_UNION = _SHARD + r'''
out = sys.argv[1]
res = {}
for r in (0, 1):
  s, st = run_shard(r, 0)
  res['local%%d' %% r] = s.cpu().numpy(); res['state%%d' %% r] = st
np.savez(out, **res)
'''


def test_two_gpu_shards_and_rccl_stats_all_reduce(tmp_path):
  import torch
  if torch.cuda.device_count() < 2:
    pytest.skip('needs >= 2 GPUs (one process per GPU over RCCL); this box has %d' % torch.cuda.device_count())
  fmt = {'root': ROOT, 'n': N_PER_RANK, 'steps': STEPS, 'max_steps': MAX_STEPS}
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  procs, outs = [], []
  for r in range(2):
    env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    outs.append(str(tmp_path / ('rank%d.npz' % r)))
    procs.append(subprocess.Popen([sys.executable, '-c', _RANK % fmt, outs[-1]], env=env))
  try:
    rcs = [p.wait(timeout=600) for p in procs]
  finally:
    for p in procs:
      if p.poll() is None:
        p.kill()
  assert rcs == [0, 0]
  union_out = str(tmp_path / 'union.npz')
  subprocess.run([sys.executable, '-c', _UNION % fmt, union_out], check=True, timeout=600)
  r0, r1, u = np.load(outs[0]), np.load(outs[1]), np.load(union_out)
  assert str(r0['backend']) == 'nccl'
  # every rank holds the same total, and it is the sum of the two shards
  np.testing.assert_array_equal(r0['total'], r1['total'])
  np.testing.assert_allclose(r0['total'], r0['local'] + r1['local'], rtol=1e-15)
  episodes = N_PER_RANK * (STEPS // (MAX_STEPS + 1))
  assert r0['local'][2] == r1['local'][2] == episodes and r0['local'][0] != r1['local'][0]
  # ... and equals the same two shards stepped by ONE process on one GPU: robots are independent, the GPUs are
  # the same hardware (states bit-identical; the statistics are double atomics in scheduling order)
  np.testing.assert_array_equal(r0['state'], u['state0'])
  np.testing.assert_array_equal(r1['state'], u['state1'])
  np.testing.assert_allclose(r0['total'], u['local0'] + u['local1'], rtol=1e-12)
