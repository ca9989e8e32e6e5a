"""N > 1 path on the CPU: two gloo ranks, each stepping its own shard of robots (emulator-backed
engine, different action seeds), then ONE all-reduce of the episodic-return statistics — the
only collective of the design.  The reduced statistics must equal the union of
the shards computed in a single process."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
STEPS, MAX_STEPS, N_PER_RANK = 8, 3, 2


def run_shard(rank):
  sys.path.insert(0, HERE)
  from test_env_host import EmuSolo8VanillaEnv
  from gym_solo_amd.distributed import rank_seed
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig
  from gym_solo_amd.workloads import register_benchmark_workload
  cfg = Solo8VanillaConfig()
  cfg.dtype, cfg.num_envs, cfg.auto_reset, cfg.settle_steps = 'float64', N_PER_RANK, True, 120
  env = EmuSolo8VanillaEnv(config=cfg)
  register_benchmark_workload(env, max_steps=MAX_STEPS)
  g = torch.Generator().manual_seed(rank_seed(1234, rank))
  for _ in range(STEPS):
    a = (torch.rand(N_PER_RANK, 12, generator=g, dtype=torch.float64) * 2 - 1) * (2 * np.pi)
    env.step(a)
  return env.engine.stats.clone()


def worker(rank, world, port, out):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                    WORLD_SIZE=str(world))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from gym_solo_amd.distributed import all_reduce_stats, rank_world
  assert rank_world() == (rank, world)
  local = run_shard(rank)
  total = all_reduce_stats(local)
  out[rank] = (local.numpy().copy(), total.numpy().copy())
  dist.barrier()
  dist.destroy_process_group()


def test_two_rank_stats_all_reduce():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  mgr = mp.Manager()
  out = mgr.dict()
  mp.spawn(worker, args=(2, port, out), nprocs=2, join=True)
  (l0, t0), (l1, t1) = out[0], out[1]
  np.testing.assert_allclose(t0, t1, rtol=0, atol=0)
  np.testing.assert_allclose(t0, l0 + l1, rtol=1e-15)
  # each shard finished STEPS // (MAX_STEPS + 1) episodes per robot, with different returns
  assert l0[2] == l1[2] == N_PER_RANK * (STEPS // (MAX_STEPS + 1))
  assert l0[0] != l1[0]
  # and equals the same two shards run in this process
  ref = run_shard(0).numpy() + run_shard(1).numpy()
  np.testing.assert_allclose(t0, ref, rtol=1e-13)
  from gym_solo_amd.distributed import shard_sizes, summarize
  assert shard_sizes(10, 4) == [3, 3, 2, 2] and shard_sizes(8192, 2) == [4096, 4096]
  s = summarize(t0)
  assert s['episodes'] == t0[2] and s['mean_length'] == MAX_STEPS + 1
