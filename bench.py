#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched Solo8 hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU.  A "step" is one pass of the whole hot path over one batch of 4096 robots
per GPU (BASELINE.json configs[1]): fresh random actions -> action
de-normalisation -> POSITION_CONTROL motors -> articulated forward dynamics -> ground contact PGS
-> integration -> TorsoIMU+MotorEncoder observations (21 floats) -> the examples' stand reward
-> TimeBasedTermination(1000) with auto-reset, all in ONE fused kernel through the C-ABI
(solo_engine_rollout: open-loop, --steps-per-launch consecutive steps of each robot per launch,
the batch cut into --rollout-streams independent launch chains).  The action pool is generated on the device before the timed
region.  The env batch is sharded over ranks with no data-path collective; the only
communication is one RCCL all-reduce of the 8-double episodic-return statistics vector at the
end of the interval (inside the timed region).

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

BYTES_PER_ENV_STEP = {'float32': 385, 'float64': 765}  # SURVEY.md §8d algorithmic bytes
HBM_PEAK_GBPS = 8000.0                                  # MI355X_MICROARCH.md chip table
NUM_SIMDS = 1024                                        # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
VALU_ISSUE_CYCLES = 4                                   # one wave64 VALU instruction occupies its SIMD 4 cycles
SHADER_CLOCK_HZ = 2.4e9                                 # max clock, MI355X_MICROARCH.md chip table (matches the
                                                        # in-kernel s_memtime rate measured with tools/gpu_stamps.py)


def build_env(num_envs, device, dtype, max_steps=1000, steps_per_launch=1, rollout_streams=1):
  import numpy as np
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig, Solo8VanillaEnv
  from gym_solo_amd.workloads import register_benchmark_workload
  cfg = Solo8VanillaConfig()
  cfg.num_envs, cfg.device, cfg.dtype, cfg.auto_reset = num_envs, device, dtype, True
  cfg.steps_per_launch, cfg.rollout_streams = steps_per_launch, rollout_streams
  env = Solo8VanillaEnv(config=cfg, copy_outputs=False)
  register_benchmark_workload(env, max_steps=max_steps)
  env._ensure_program()
  return env


def host_cores():
  """Cores this process may actually use: the GPU box exposes 256 logical CPUs but gives a
  1-GPU job a 16-CPU share (cgroup quota), and oversubscribed OpenMP threads crawl."""
  n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
  try:
    quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
    if quota != 'max':
      n = min(n, max(1, int(int(quota) / int(period))))
  except Exception:  # noqa: BLE001
    pass
  return max(1, min(n, int(os.environ.get('SOLO_CPU_BASELINE_THREADS', '16'))))


def cpu_baseline(num_envs, seconds_target=12.0):
  """The CPU oracle (double-precision scalar C restatement + numpy reductions) on the host
  cores, bounded sample of the same workload.  kind = "port": PyBullet itself is not
  installable in this pipeline."""
  import subprocess
  import tempfile
  import numpy as np
  from helpers import make_abi
  from oracle import solo_oracle as so
  import env_cases
  cores = host_cores()
  tmp = tempfile.mkdtemp(prefix='solo_oracle_native_')
  lib_path = os.path.join(tmp, 'libsolo_oracle_native.so')
  try:
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle'), 'native', 'OUT=' + lib_path])
  except Exception:  # noqa: BLE001 — fall back to the portable build
    lib_path = None
  ca, ma = make_abi('float64', auto_reset=True)
  env = so.OracleEnv(ca, ma, num_envs, [('torso_imu', {}), ('motor_encoder', {})],
                     [(1, env_cases.BENCH_REWARD)], [('time', 1000)], threads=cores)
  if lib_path:
    env.phys = so.OraclePhysics(ca, ma, lib_path)
  rng = np.random.default_rng(1234)
  env.step(rng.uniform(-2 * np.pi, 2 * np.pi, (num_envs, 12)))  # warm-up (thread pool, TLS)
  steps, t0 = 0, time.perf_counter()
  while True:
    env.step(rng.uniform(-2 * np.pi, 2 * np.pi, (num_envs, 12)))
    steps += 1
    el = time.perf_counter() - t0
    if el > seconds_target or steps >= 2000:
      break
  # the same oracle on ONE core (a short sample), for the per-core figure SURVEY.md §8d asks for
  env1 = so.OracleEnv(ca, ma, 256, [('torso_imu', {}), ('motor_encoder', {})],
                      [(1, env_cases.BENCH_REWARD)], [('time', 1000)], threads=1)
  if lib_path:
    env1.phys = so.OraclePhysics(ca, ma, lib_path)
  env1.step(rng.uniform(-2 * np.pi, 2 * np.pi, (256, 12)))
  s1, t1 = 0, time.perf_counter()
  while time.perf_counter() - t1 < 3.0:
    env1.step(rng.uniform(-2 * np.pi, 2 * np.pi, (256, 12)))
    s1 += 1
  one_core = 256 * s1 / (time.perf_counter() - t1)
  return {'value': num_envs * steps / el, 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port', 'value_one_core': one_core,
          'sample': '%d envs x %d steps of the same workload on the f64 C oracle (OpenMP over envs, '
                    '%d threads) + numpy obs/reward, %.1f s' % (num_envs, steps, cores, el)}


def pmc_profile(dtype, key):
  """A per-launch / per-env-step figure from the committed rocprofv3 --pmc run (profiles/), or None."""
  path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
  try:
    with open(path) as f:
      return json.load(f).get(dtype, {}).get(key)
  except Exception:  # noqa: BLE001
    return None


def secondary_bound(dtype, env_steps_per_launch, chains, kern_ms, clock_hz):
  """The bound that actually binds (the path is VALU-issue / latency bound, not
  HBM bound): share of the chip's VALU issue slots the launches keep busy, from the VALU
  instruction count per env-step measured with rocprofv3 --pmc SQ_INSTS_VALU (profiles/)."""
  valu = pmc_profile(dtype, 'valu_insts_per_env_step')
  if not valu or not clock_hz:
    return 'VALU-issue / latency bound by construction (SURVEY.md §8d); no PMC profile committed for this dtype'
  simd_cycles = NUM_SIMDS * kern_ms * 1e-3 * clock_hz
  util = chains * valu * env_steps_per_launch * VALU_ISSUE_CYCLES / simd_cycles
  return ('VALU-issue / latency bound by construction (SURVEY.md §8d), not HBM bound: %.0f VALU instructions per '
          'env-step (rocprofv3 --pmc SQ_INSTS_VALU, profiles/pmc_traffic.json) x %d cycles x %d env-steps x %d '
          'concurrent launch chains = %.2f of the %d SIMDs\' issue cycles over the measured launch duration at %.2f GHz'
          % (valu, VALU_ISSUE_CYCLES, env_steps_per_launch, chains, util, NUM_SIMDS, clock_hz / 1e9))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=3000)
  ap.add_argument('--warmup', type=int, default=250)
  ap.add_argument('--envs-per-gpu', type=int, default=4096)
  ap.add_argument('--dtype', default='float32', choices=['float32', 'float64'])
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--steps-per-launch', type=int, default=250,
                  help='env steps of every robot fused into one kernel launch by the open-loop rollout '
                       '(1 = one launch per step, the closed-loop granularity)')
  ap.add_argument('--rollout-streams', type=int, default=2,
                  help='batch slices advancing as independent launch chains on separate HIP streams')
  ap.add_argument('--api-rate', action='store_true',
                  help='also time Solo8VanillaEnv.step() in a python loop (one launch per step)')
  args = ap.parse_args()

  import torch
  import torch.distributed as dist
  from gym_solo_amd import abi
  from gym_solo_amd.distributed import all_reduce_stats, rank_seed, summarize

  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  if world != args.gpus:
    raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run' % (args.gpus, world))
  if not torch.cuda.is_available():
    raise SystemExit('bench.py needs an MI355X: the engine has no CPU fallback')
  torch.cuda.set_device(local_rank)
  # under torch.distributed.run the collective path is exercised even with one rank
  distributed = world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ or os.environ.get('SOLO_BENCH_FORCE_DIST') == '1'
  if distributed:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', rank=rank, world_size=world,
                            device_id=torch.device('cuda', local_rank))

  n, k, w = args.envs_per_gpu, args.steps, args.warmup
  tdtype = torch.float32 if args.dtype == 'float32' else torch.float64
  # (a run shorter than one fused launch fuses what it has: K steps per launch)
  spl, streams = max(1, min(args.steps_per_launch, k)), max(1, args.rollout_streams)
  env = build_env(n, local_rank, args.dtype, steps_per_launch=spl, rollout_streams=streams)
  eng = env.engine
  gen = torch.Generator(device='cuda:%d' % local_rank).manual_seed(rank_seed(1234, rank))
  two_pi = 2 * 3.141592653589793

  def action_pool(steps):
    a = torch.rand(steps, n, abi.NUM_JOINTS, device='cuda:%d' % local_rank, dtype=tdtype, generator=gen)
    return (a * 2 - 1) * two_pi

  def barrier():
    if distributed:
      dist.barrier()
    torch.cuda.synchronize(local_rank)

  if w > 0:
    eng.rollout(action_pool(w), abi.STEP_ALL)
  acts = action_pool(k)
  out = eng.rollout_buffers(k)  # every step's obs / reward / done is written out (to HBM)
  stats_before = eng.stats.clone()
  barrier()
  t0 = time.perf_counter()
  eng.rollout(acts, abi.STEP_ALL, out=out)
  # the ONLY collective: episodic-return statistics, 64 B, RCCL over xGMI
  stats = all_reduce_stats(eng.stats - stats_before)
  barrier()  # (all_reduce_stats is a no-op without an initialised process group)
  elapsed = time.perf_counter() - t0
  t = torch.tensor([elapsed], dtype=torch.float64, device='cuda:%d' % local_rank)
  if distributed:
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
  elapsed = float(t.item())

  # dominant kernel: HIP events on the stream its launches are issued on, same rollout path and
  # workload (fresh actions every step); one launch = (n / streams) robots x spl steps
  reps = max(1, min(k, 1000) // spl)
  kern_ms = eng.time_step(acts[:reps * spl], abi.STEP_ALL)
  api_rate = None
  if args.api_rate:  # API-level rate through Solo8VanillaEnv.step (python loop, zero-copy outputs)
    api_steps = min(k, 200)
    torch.cuda.synchronize(local_rank)
    ta = time.perf_counter()
    for i in range(api_steps):
      env.step(acts[i])
    torch.cuda.synchronize(local_rank)
    api_rate = n * api_steps / (time.perf_counter() - ta)

  st = stats.cpu().numpy()
  if rank == 0:
    value = world * n * k / elapsed
    robots_per_launch = n // streams if (streams > 1 and n >= 2 * streams and k > 1) else n
    env_steps_per_launch = robots_per_launch * spl
    bytes_per_launch = BYTES_PER_ENV_STEP[args.dtype] * env_steps_per_launch
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
    line = {
      'metric': 'env-steps/s (whole node), 4096 Solo8 envs/GPU, 1/2/4/8 MI355X', 'value': value, 'unit': 'env-steps/s',
      'n_gpus': world, 'steps': k, 'warmup': w, 'ms_per_step': elapsed / k * 1e3,
      'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
      'dtype': 'f32' if args.dtype == 'float32' else 'f64', 'data': 'synthetic',
      'config': {'workload': 'BASELINE configs[1]: %d Solo8 envs/GPU, flat ground, POSITION_CONTROL, '
                             'U(-2pi,2pi) actions, TorsoIMU+MotorEncoder obs, stand reward, '
                             'TimeBasedTermination(1000)+auto-reset, dt=1e-3, 50 PGS iterations' % n,
                 'envs_per_gpu': n, 'steps_per_launch': spl, 'rollout_streams': streams, 'parallelism': 'env-batch sharded x%d, RCCL all-reduce of return stats only' % world},
      'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                   'frac': achieved / HBM_PEAK_GBPS, 'traffic': pmc_profile(args.dtype, 'hbm_bytes_per_launch'),
                   'kernel': eng.kernel_name, 'kernel_ms': kern_ms,
                   'algorithmic_bytes_per_launch': bytes_per_launch, 'env_steps_per_launch': env_steps_per_launch,
                   'concurrent_launch_chains': n // robots_per_launch,
                   'achieved_all_chains': achieved * (n // robots_per_launch),
                   'note': secondary_bound(args.dtype, env_steps_per_launch, n // robots_per_launch, kern_ms,
                                           SHADER_CLOCK_HZ)},
      'episodes': summarize(st),
      'env_api_env_steps_per_s_rank0': api_rate,
    }
    if world == 1 and not args.no_cpu_baseline:
      line['cpu_baseline'] = cpu_baseline(n)
    else:
      line['cpu_baseline'] = None
    print(json.dumps(line), flush=True)
  if distributed:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
