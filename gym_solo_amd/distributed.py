"""Multi-GPU plumbing: one process per GPU, the env batch sharded across ranks.

The hot path never communicates: robots are independent (the reference gives every env its own
private BulletClient, gym_solo/envs/solo8_base_env.py:34; README.md:25), so each rank steps its
own shard.  The ONLY collective is a sum all-reduce of the 8-double episodic-return statistics
vector (RCCL over xGMI when the backend is "nccl"; latency-bound, 64 bytes), once per reporting
interval.
"""
import math
import os


def rank_world():
  return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


def rank_seed(base_seed: int, rank: int) -> int:
  """Action-stream seed of a rank (1234 + rank)."""
  return int(base_seed) + int(rank)


def shard_sizes(total_envs: int, world: int):
  """Split `total_envs` robots over `world` ranks as evenly as possible (weak scaling uses
  total = per_gpu * world, where every shard is per_gpu)."""
  base, rem = divmod(int(total_envs), int(world))
  return [base + (1 if r < rem else 0) for r in range(world)]


def all_reduce_stats(stats, group=None, in_place=False):
  """Sum the per-rank statistics [sum return, sum return^2, episodes, sum length, -, diverged, migration waits given up, -]
  over all ranks.  Returns a new tensor (or `stats` itself with in_place=True); a no-op without an
  initialised process group."""
  import torch.distributed as dist
  out = stats if in_place else stats.clone()
  if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
  return out


def summarize(stats) -> dict:
  s = [float(x) for x in stats]
  n = s[2]
  if n <= 0:
    return {'episodes': 0.0, 'mean_return': None, 'std_return': None, 'mean_length': None,
            'diverged': s[5], 'migration_waits_given_up': s[6]}
  mean = s[0] / n
  var = max(0.0, s[1] / n - mean * mean)
  # (slot 6: waves of a migrating launch that gave up waiting for a ring slot - a launch must never hang on a bug; always 0)
  return {'episodes': n, 'mean_return': mean, 'std_return': math.sqrt(var),
          'mean_length': s[3] / n, 'diverged': s[5], 'migration_waits_given_up': s[6]}
