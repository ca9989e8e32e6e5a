#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched Solo8 hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One process per GPU.  With N > 1 and no torch.distributed environment the script launches its own N
ranks (child processes, started BEFORE anything touches the GPU; the parent only waits and forwards
rank 0's JSON line); under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
it uses the ranks it is given.

A "step" is one pass of the whole hot path over one batch of 4096 robots per GPU (BASELINE.json
configs[1]): fresh random actions -> action de-normalisation -> POSITION_CONTROL
motors -> articulated forward dynamics -> ground contact PGS -> integration -> TorsoIMU +
MotorEncoder observations (21 floats) -> the examples' stand reward -> TimeBasedTermination(1000)
with auto-reset, through the C-ABI (solo_engine_rollout_record: every step's obs / reward / done is
written to HBM; --steps-per-launch consecutive steps of each robot per kernel launch).  Actions are
generated on the device before the timed region.  The env batch is sharded over ranks with no
data-path collective; the only communication is one RCCL all-reduce of the 8-double
episodic-return statistics vector at the end of the interval (inside the timed region).

Timing: after W warm-up steps, EXACTLY K steps are timed between barrier + device synchronisation
on both sides, max over ranks.  A K-step region can be as short as half a millisecond (the driver
uses K = 20), so it is repeated (fresh actions, the simulation simply continues) until 2 s (or 4000
repeats) have accumulated; `value` / `ms_per_step` are the MEDIAN repeat, min / max / count are
reported next to it.  Rank 0 prints ONE JSON line with `roofline` and `cpu_baseline`; the same run
also measures the reference-precision figure (`value_f64`) and the reference-granularity figure
(`value_closed_loop`: one solo_engine_step launch per env step, as Solo8VanillaEnv.step issues it).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_ENV_STEP = {'float32': 385, 'float64': 765}  # SURVEY.md §8d algorithmic bytes
HBM_PEAK_GBPS = 8000.0                                  # MI355X_MICROARCH.md chip table
NUM_SIMDS = 1024                                        # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
VALU_CYCLES_SHARED = 2                                  # wave64 VALU instruction on a SIMD-32 when >= 2 waves share
VALU_CYCLES_ALONE = 4                                   # the SIMD; 4 for one wave alone (MI355X_MICROARCH.md constants)
# f64 arithmetic issues at half the f32 rate (MI355X: 78.6 vs 157.3 vector TFLOP/s -> 4 cycles per wave64 instruction); 61 %
# of the f64 step kernel's VALU instructions are f64 arithmetic (static count of the product assembly, tools/isa_line_profile.py d:
# 1845 of 3162 in round 5's 128-VGPR kernel), the rest - moves, DPP, selects, integer, compares on 32-bit halves - 32-bit operations
F64_ARITH_SHARE = 0.58
SHADER_CLOCK_HZ = 2.4e9                                 # max clock, MI355X_MICROARCH.md chip table
KERNEL_MS_SAMPLES = int(os.environ.get('SOLO_BENCH_KERNEL_SAMPLES', '101'))   # HIP-event samples of the dominant launch (roofline.kernel_ms: their median)
METRIC = 'env-steps/s (whole node), 4096 Solo8 envs/GPU, 1/2/4/8 MI355X'
# REHEARSAL knobs (never set by the driver; tests/test_bench_launcher.py): SOLO_BENCH_ENGINE=emu runs the whole script -
# launcher, process group (gloo), timed regions, statistics all-reduce, JSON line - on the CPU wave emulator of the
# product kernel source, a few robots per rank; SOLO_BENCH_MAX_STEPS shortens the episodes so that a tiny run ends some.
EMU = os.environ.get('SOLO_BENCH_ENGINE') == 'emu'
MAX_STEPS = int(os.environ.get('SOLO_BENCH_MAX_STEPS', '1000'))


def build_env(num_envs, device, dtype, max_steps=MAX_STEPS, steps_per_launch=-1, rollout_streams=-1, residual_threshold=0.0, migrate_steps=-1,
              warm_start=0.0, ulp_tolerance=None):
  """-1 for the three launch knobs = the engine chooses (SoloConfig's defaults: the measured launch policy lives in the
  engine since round 5, Engine.plan(k) reports it)."""
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig, Solo8VanillaEnv
  from gym_solo_amd.workloads import register_benchmark_workload
  if EMU:  # CPU REHEARSAL of the launcher / collective / JSON plumbing (tests/test_bench_launcher.py): never a measurement
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from emu_kernel import make_emu_env_class
    Solo8VanillaEnv = make_emu_env_class()
  cfg = Solo8VanillaConfig()
  cfg.num_envs, cfg.device, cfg.dtype, cfg.auto_reset = num_envs, device, dtype, True
  cfg.steps_per_launch, cfg.rollout_streams = steps_per_launch, rollout_streams
  cfg.solver_residual_threshold = residual_threshold
  cfg.solver_warm_start = warm_start
  cfg.migrate_steps = migrate_steps
  cfg.solver_ulp_tolerance = ulp_tolerance   # (None: the precision's default - 512 half-ulps in f64, 2 in f32: core/configs.py)
  if EMU:
    cfg.settle_steps = 30
  env = Solo8VanillaEnv(config=cfg, copy_outputs=False)
  register_benchmark_workload(env, max_steps=max_steps)
  env._ensure_program()
  return env


def desynchronise_episodes(eng, generator, max_steps=MAX_STEPS, advance=True, chunk=None):
  """Brings the batch into the STEADY STATE of the workload before anything is timed.  All robots are created
  in the same post-reset pose with their 1000-step episodes in phase: a 20-step window right after that would
  never see a termination, an auto-reset or a real number in the statistics all-reduce, and would time 4096
  copies of the first steps of an episode (folded on the ground, every knee and foot in contact) instead of
  the mix of episode phases an RL run is in.  So every robot's TimeBased counter is seeded uniformly in
  [0, max_steps) and the simulation is advanced by max_steps untimed steps of the same random-action workload
  with the in-kernel auto-reset: each robot ends its (shortened) first episode at its own time and is
  afterwards `phase` steps into a regular episode - counters AND physical states spread uniformly over the
  episode.  ~K/1000 of the robots then finish an episode inside any K-step window."""
  import torch
  from gym_solo_amd import abi
  n, dev = eng.num_envs, eng.state.device
  phase = torch.randint(0, max_steps, (n,), device=dev, generator=generator, dtype=torch.int32)
  # the TimeBased termination's counter: its slot in the registered program (not "column 0")
  prog = eng.program
  slots = [i for i in range(prog.num_terms) if prog.term_kind[i] == abi.T_TIME] if prog is not None else []
  if len(slots) != 1:
    raise SystemExit('desynchronise_episodes expects exactly one TimeBasedTermination in the workload, found %d' % len(slots))
  eng.term_count[:, slots[0]] = phase
  eng.state[:, abi.S_EPLEN] = phase.to(eng.tdtype)
  if not advance:
    return
  # (chunk: steps per untimed rollout - bench.py passes the length whose launches have the geometry of the timed region, so
  # that a profile of the run holds ONE kind of step-kernel launch and its --stats average compares with roofline.kernel_ms)
  chunk = min(chunk or 100, max_steps)
  for _ in range(0, max_steps, chunk):
    a = (torch.rand(chunk, n, abi.NUM_JOINTS, device=dev, dtype=eng.tdtype, generator=generator) * 2 - 1) * (2 * 3.141592653589793)
    eng.rollout(a, abi.STEP_ALL)
  if dev.type == 'cuda':
    torch.cuda.synchronize(dev)


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


def cpu_baseline(num_envs, seconds_target=float(os.environ.get('SOLO_CPU_BASELINE_SECONDS', '12'))):
  """The CPU oracle (double-precision scalar C restatement + numpy reductions) on the host
  cores, bounded sample of the same workload.  kind = "port": PyBullet itself is not
  installable in this pipeline.  The ONLY place bench.py touches tests/ or
  oracle/ - as the measured CPU baseline, never inside the GPU path."""
  import tempfile
  import numpy as np
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
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
  while time.perf_counter() - t1 < min(3.0, seconds_target):
    env1.step(rng.uniform(-2 * np.pi, 2 * np.pi, (256, 12)))
    s1 += 1
  one_core = 256 * s1 / (time.perf_counter() - t1)
  return {'value': num_envs * steps / el, 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port', 'value_one_core': one_core,
          'sample': '%d envs x %d steps of the same workload on the f64 C oracle (OpenMP over envs, '
                    '%d threads) + numpy obs/reward, %.1f s' % (num_envs, steps, cores, el)}


def pybullet_baseline(seconds_target=5.0):
  """BASELINE.md §4 (opportunistic): if pybullet happens to be importable on this box, time BASELINE
  configs[0] - ONE Solo8 on PyBullet DIRECT, random actions, the reference's own call sequence
  (solo8v2vanilla.py:87-91) on a URDF written from the build's model constants.  Never the case in
  this pipeline (no network, only the repo travels); then the field says so and no number is quoted."""
  try:
    import pybullet as p
    import pybullet_data
  except Exception:  # noqa: BLE001
    return {'available': False, 'note': 'PyBullet unavailable - reference CPU timing not measured'}
  try:
    import tempfile
    import numpy as np
    from gym_solo_amd.urdf import to_urdf
    path = os.path.join(tempfile.mkdtemp(prefix='solo_urdf_'), 'solo.urdf')
    with open(path, 'w') as f:
      f.write(to_urdf())
    cid = p.connect(p.DIRECT)
    p.setAdditionalSearchPath(pybullet_data.getDataPath(), physicsClientId=cid)
    p.setGravity(0, 0, -9.81, physicsClientId=cid)
    p.setPhysicsEngineParameter(fixedTimeStep=1e-3, numSubSteps=1, physicsClientId=cid)
    p.loadURDF('plane.urdf', physicsClientId=cid)
    robot = p.loadURDF(path, [0, 0, 0.5], p.getQuaternionFromEuler([0, 0, 0]), flags=p.URDF_USE_INERTIA_FROM_FILE,
                       useFixedBase=False, physicsClientId=cid)
    n_j = p.getNumJoints(robot, physicsClientId=cid)
    rng = np.random.default_rng(1234)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_target:
      a = rng.uniform(-2 * np.pi, 2 * np.pi, n_j)
      p.setJointMotorControlArray(robot, np.arange(n_j), p.POSITION_CONTROL, targetPositions=a, forces=[2.0] * n_j,
                                  physicsClientId=cid)
      p.stepSimulation(physicsClientId=cid)
      steps += 1
    el = time.perf_counter() - t0
    p.disconnect(cid)
    return {'available': True, 'value': steps / el, 'unit': 'env-steps/s', 'cores': 1,
            'sample': '1 Solo8 on PyBullet DIRECT (%s), physics calls only, %d steps' % (getattr(p, '__version__', '?'), steps)}
  except Exception as e:  # noqa: BLE001
    return {'available': False, 'note': 'pybullet importable but the run failed: %r' % (e,)}


def pmc_profile(dtype, spl, slices):
  """Per-env-step figures from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json, written by
  tools/make_pmc_traffic.py), or {}.  The file holds one entry per dtype and launch geometry
  ("float32", "float32_k20", "float64", "float64_k20": fused launches of 250 steps on two stream slices, or
  the driver's single 20-step launch); the entry measured on THIS run's geometry is used when there is one,
  else the dtype's other entry - `geometry_match` says which."""
  try:
    with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
      table = json.load(f)
  except Exception:  # noqa: BLE001
    return {}
  # a profile is only as good as the binary it was taken on: every entry carries the hash of the kernel sources
  # (gym_solo_amd/build_info.py); an entry from other sources is reported as stale and none of its figures is quoted
  from gym_solo_amd.build_info import kernel_source_hash
  now = kernel_source_hash()
  def pick(key, e, match):
    if e.get('kernel_source_hash') != now:
      return {'stale': True, 'profile_key': key, 'profile_hash': e.get('kernel_source_hash'), 'source_hash': now}
    return dict(e, geometry_match=match, profile_key=key, stale=False, source_hash=now)
  for key, e in table.items():
    if key.startswith(dtype) and e.get('steps_per_launch') == spl and e.get('launch_chains') == slices:
      return pick(key, e, True)
  for key in (dtype + '_k20' if spl <= 20 else dtype, dtype, dtype + '_k20'):
    if key in table:
      return pick(key, table[key], False)
  return {}


def secondary_bound(pmc, env_steps_per_launch, chains, kern_ms, waves_per_simd, dtype='float32'):
  """The bound that actually binds (the path is bound by the instruction issue rate of one
  wave per robot, not by HBM): share of the chip's VALU issue capacity the launches use, from the VALU
  instruction count per env-step measured with rocprofv3 --pmc SQ_INSTS_VALU (profiles/)."""
  valu = pmc.get('valu_insts_per_env_step')
  if not valu:
    return 'dependent-issue-latency bound by construction (SURVEY.md §8d); no PMC profile committed for this dtype', None
  # MEASURED (quote the counter, not a model): SQ_ACTIVE_INST_VALU - quad-cycles a SIMD's VALU was executing, summed
  # over the waves - x 4 / (SIMDs x the launch's cycles, GRBM_GUI_ACTIVE / 8 XCDs), from the same --pmc pass of the same kernel
  # sources and launch geometry (tools/make_pmc_traffic.py)
  measured = pmc.get('valu_busy_share_measured')
  if measured is None and pmc.get('step_kernel_wave_cycle_shares') and pmc.get('grbm_gui_active_per_launch'):
    measured = (pmc['step_kernel_wave_cycle_shares']['active_inst_valu'] * pmc['step_kernel_wave_cycles'] * 4.0 /
                (NUM_SIMDS * pmc['grbm_gui_active_per_launch'] / 8.0))
  cyc = VALU_CYCLES_SHARED if waves_per_simd >= 2 else VALU_CYCLES_ALONE
  if dtype == 'float64':  # (the mean over the kernel's mix of f64 arithmetic and 32-bit operations)
    cyc = cyc * (1.0 + F64_ARITH_SHARE)
  clock = pmc.get('effective_clock_hz') or SHADER_CLOCK_HZ  # measured (GRBM_GUI_ACTIVE / 8 / wall) when profiled
  simd_cycles = NUM_SIMDS * kern_ms * 1e-3 * clock
  util = chains * valu * env_steps_per_launch * cyc / simd_cycles
  return ('latency bound, not HBM bound: %.0f VALU instructions per env-step (rocprofv3 --pmc SQ_INSTS_VALU, '
          'profiles/pmc_traffic.json) x %.1f cycles (wave64 on a SIMD-32 shared by %.0f waves, MI355X_MICROARCH.md; f64 arithmetic - 58 %% of the f64 kernel\'s VALU instructions - at half rate) x %d '
          'env-steps x %d concurrent launch chains = %.2f of the %d SIMDs\' VALU issue capacity over the measured launch '
          'duration at %.1f GHz; what binds is the issue rate of ONE wave per robot - 4.1 cycles per independent instruction, '
          '6.2 per instruction of the solver\'s serial row-update chain (tools/microbench/simd_rate.hip), %s instructions per '
          'env-step - and the launch waiting for its slowest robot (DESIGN.md section 4)'
          % (valu, cyc, waves_per_simd, env_steps_per_launch, chains, util, NUM_SIMDS, clock / 1e9,
             ('%.0f' % pmc['insts_per_env_step']) if pmc.get('insts_per_env_step') else '~2300'),
          {'bound': 'valu-issue', 'frac': measured if measured is not None else util, 'frac_source': 'measured: SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) of the committed --pmc pass' if measured is not None else 'modelled (no counter in the profile)',
           'frac_modelled': util, 'wave_cycle_shares': pmc.get('step_kernel_wave_cycle_shares'),
           'valu_insts_per_env_step': valu, 'insts_per_env_step': pmc.get('insts_per_env_step'),
           'cycles_per_valu_inst': cyc, 'simds': NUM_SIMDS, 'clock_hz': clock, 'concurrent_launch_chains': chains,
           'unit': 'share of the SIMDs\' VALU issue slots over the launch (SURVEY.md §8d: the secondary, practical bound)'})


def free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def launch_ranks(n_ranks, child_cmd=None):
  """N > 1 without a torch.distributed environment: start one child per GPU (this process has not
  touched the GPU and never will), forward rank 0's stdout, exit with the worst child status.
  (child_cmd: the command every rank runs - this script with its own arguments unless a test passes another.)"""
  port = os.environ.get('MASTER_PORT') or str(free_port())
  procs = []
  for r in range(n_ranks):
    env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR='127.0.0.1',
               MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
               SOLO_BENCH_CHILD='1')
    procs.append(subprocess.Popen(child_cmd or [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                  stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
  # rank 0's stdout is drained by a thread while ALL children are polled: the first rank that exits
  # non-zero (GPU unavailable, import error) takes the others down at once instead of leaving them - and
  # this parent - waiting in init_process_group / a barrier until the store or RCCL timeout
  import threading
  chunks = []
  reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
  reader.start()
  failed = None
  while failed is None and any(p.poll() is None for p in procs):
    for r, p in enumerate(procs):
      if p.poll() not in (None, 0):
        failed = r
        break
    time.sleep(0.2)
  if failed is None:
    failed = next((r for r, p in enumerate(procs) if p.returncode != 0), None)
  if failed is not None:
    print('[bench launcher] rank %d exited with status %d: stopping the other ranks' % (failed, procs[failed].returncode),
          file=sys.stderr, flush=True)
    for p in procs:
      if p.poll() is None:
        p.terminate()
    for p in procs:
      try:
        p.wait(timeout=20)
      except subprocess.TimeoutExpired:
        p.kill()
  rcs = [p.wait() for p in procs]
  reader.join(timeout=10)
  sys.stdout.write(b''.join(c for c in chunks if c).decode())
  sys.stdout.flush()
  raise SystemExit(max(abs(rc) for rc in rcs))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=3000)
  ap.add_argument('--warmup', type=int, default=250)
  ap.add_argument('--envs-per-gpu', type=int, default=4096)
  ap.add_argument('--dtype', default='float64', choices=['float32', 'float64'],
                  help='arithmetic of the headline `value`: float64 is the reference\'s precision (PyBullet computes in double: '
                       'solo8v2vanilla.py:91) and the default; float32 is the opt-in fast mode, reported next to it as value_f32')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-extra', action='store_true', help='skip the value_f64 / value_closed_loop legs')
  ap.add_argument('--steps-per-launch', type=int, default=-1,
                  help='env steps of every robot fused into one kernel launch by the open-loop rollout '
                       '(1 = one launch per step, the closed-loop granularity; -1, the default = the engine chooses: min(K, 250))')
  ap.add_argument('--rollout-streams', type=int, default=-1,
                  help='batch slices advancing as independent launch chains on separate HIP streams '
                       '(-1, the default = the engine chooses: two when a rollout takes several launches)')
  ap.add_argument('--migrate-steps', type=int, default=-1,
                  help='SoloConfig.migrate_steps of the rollouts: robots change waves every this many steps of a launch '
                       '(0 = off; -1, the default = the engine chooses: only when a launch has more robots than the chip has wave slots)')
  ap.add_argument('--min-seconds', type=float, default=2.0, help='repeat the K-step timed region until this much time ... (round 5: 0.5)')
  ap.add_argument('--max-repeats', type=int, default=4000, help='... or this many repeats have accumulated (round 4: 30 - twenty milliseconds of GPU work at K = 20; round 5: 1000)')
  args = ap.parse_args()

  world = int(os.environ.get('WORLD_SIZE', '1'))
  if args.gpus > 1 and world == 1 and 'RANK' not in os.environ:
    launch_ranks(args.gpus)  # does not return
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus:
    raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))

  import torch
  import torch.distributed as dist
  from gym_solo_amd import abi
  from gym_solo_amd.distributed import all_reduce_stats, rank_seed, summarize

  if not EMU and not torch.cuda.is_available():
    raise SystemExit('bench.py needs an MI355X: the engine has no CPU fallback')
  # rehearsal knobs for a 1-GPU box (never set by the driver): every rank on cuda:0, gloo instead of
  # RCCL (which refuses two ranks on one device) - exercises the launcher and the multi-rank timing
  backend = os.environ.get('SOLO_BENCH_BACKEND', 'gloo' if EMU else 'nccl')
  if os.environ.get('SOLO_BENCH_SHARE_GPU') == '1':
    local_rank = 0
  if not EMU:
    torch.cuda.set_device(local_rank)
  # under torch.distributed.run the collective path is exercised even with one rank
  distributed = world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ or os.environ.get('SOLO_BENCH_FORCE_DIST') == '1'
  log = (lambda *a: print('[bench rank %d]' % rank, *a, file=sys.stderr, flush=True)) if distributed else (lambda *a: None)
  if distributed:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29531')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend == 'nccl':
      dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))  # (RCCL)
    else:
      dist.init_process_group(backend, rank=rank, world_size=world)
    log("init_process_group('%s') ok: world_size %d, backend %s, device cuda:%d" % (backend, world, dist.get_backend(), local_rank))
  # every rank says what it runs on (stderr): the first multi-GPU run then shows from its tail alone that RCCL saw N ranks
  try:
    rccl = '.'.join(str(v) for v in torch.cuda.nccl.version()) if (not EMU and hasattr(torch.cuda, 'nccl')) else 'n/a'
  except Exception as e:  # noqa: BLE001
    rccl = 'unavailable (%s)' % type(e).__name__
  print('[bench rank %d] device %s, world size %d (process group: %s), RCCL %s, torch %s, HSA_ENABLE_IPC_MODE_LEGACY=%s' % (
    rank, 'cpu emulator' if EMU else torch.cuda.get_device_name(local_rank),
    dist.get_world_size() if distributed else 1, ('%s initialised' % dist.get_backend()) if distributed else 'none',
    rccl, torch.__version__, os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')), file=sys.stderr, flush=True)

  dev = 'cpu' if EMU else 'cuda:%d' % local_rank
  n, k, w = args.envs_per_gpu, args.steps, args.warmup
  two_pi = 2 * 3.141592653589793

  def barrier():
    if distributed:
      dist.barrier()
    device_sync()

  def device_sync():
    if not EMU:
      torch.cuda.synchronize(local_rank)

  def max_over_ranks(x):
    if not distributed:
      return x
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

  def timed(dtype, k, closed_loop, min_seconds, max_repeats, residual_threshold=0.0, steady=True, warm_start=0.0, ulp_tolerance=None):
    """Repeats of the K-step timed region on a fresh engine; returns per-repeat seconds (max over
    ranks), the summed episodic statistics of the timed repeats and the engine."""
    tdtype = torch.float32 if dtype == 'float32' else torch.float64
    # the launch geometry is THE ENGINE'S (SoloConfig's -1 defaults: solo_engine.hip make_plan, reported by Engine.plan(k));
    # the closed loop is one solo_engine_step launch per env step by definition; command-line values override
    env = build_env(n, local_rank, dtype, steps_per_launch=1 if closed_loop else args.steps_per_launch,
                    rollout_streams=1 if closed_loop else args.rollout_streams, residual_threshold=residual_threshold,
                    migrate_steps=0 if closed_loop else args.migrate_steps, warm_start=warm_start, ulp_tolerance=ulp_tolerance)
    eng = env.engine
    gen = torch.Generator(device=dev).manual_seed(rank_seed(1234, rank))

    def action_pool(steps):
      a = torch.rand(steps, n, abi.NUM_JOINTS, device=dev, dtype=tdtype, generator=gen)
      return (a * 2 - 1) * two_pi

    def run(acts, out):
      if closed_loop:  # one solo_engine_step launch per env step, outputs in the engine's view
        for i in range(acts.shape[0]):
          eng.step(acts[i], abi.STEP_ALL)
      else:
        eng.rollout(acts, abi.STEP_ALL, out=out)

    if steady:
      p0 = {'launches': 1} if closed_loop else eng.plan(k)
      desynchronise_episodes(eng, gen, chunk=(1 if closed_loop else k) if p0['launches'] == 1 else 2 * p0['steps_per_launch'])
    out = None if closed_loop else eng.rollout_buffers(k)  # every step's obs / reward / done goes to HBM
    if w > 0:
      run(action_pool(w), None if closed_loop else eng.rollout_buffers(w))
    def one_repeat():
      acts = action_pool(k)
      # (the interval's statistics = totals after - totals before; the totals before are reduced over the ranks here,
      # outside the timed region, the totals after inside it)
      before = all_reduce_stats(eng.stats_shards.sum(dim=0))
      barrier()
      t0 = time.perf_counter()
      run(acts, out)
      # the ONLY collective: episodic-return statistics, 64 B, RCCL over xGMI.  (One rank has nothing
      # to reduce: its statistics are read after the timed region - a reduction kernel of its own inside a 0.4-ms
      # region costs 20 us of launch and completion latency, tools/gpu_k20_segments.py.)
      if distributed:
        # ... and it IS the closing barrier: a rank's sum all-reduce completes only after every rank has contributed,
        # i.e. after every rank's K steps (stream-ordered in front of its contribution) - a dist.barrier() behind it
        # would be a second collective with the same meaning (~5 % of a 0.4-ms region on one rank)
        after = all_reduce_stats(eng.stats_shards.sum(dim=0), in_place=True)
        device_sync()
      else:
        barrier()
      t = max_over_ranks(time.perf_counter() - t0)
      if not distributed:
        after = eng.stats_shards.sum(dim=0)
      return t, after - before

    # ... and ONE untimed repeat of exactly the timed region (same K, same output buffers, the statistics
    # reduction included): first-use costs - code objects of this launch shape and of torch's small kernels,
    # page faults of the output buffers - stay out of the timed repeats
    one_repeat()
    times, total = [], None
    while True:
      t, stats = one_repeat()
      times.append(t)
      total = stats if total is None else total + stats
      if sum(times) >= min_seconds or len(times) >= max_repeats:
        break
    plan = {'steps_per_launch': 1, 'launches': k, 'slices': 1, 'migrate_steps': 0} if closed_loop else eng.plan(k)
    return times, total, eng, env, action_pool, plan

  def action_generation_ms(action_pool, k):
    """What the timed region leaves out: drawing the K x N x 12 random actions on the device - torch's generator, in
    front of the barrier.  Measured on its own (device sync on both sides, median of 21) and reported next to `value`."""
    ts = []
    for _ in range(21):
      device_sync()
      t0 = time.perf_counter()
      action_pool(k)
      device_sync()
      ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e3

  def roofline(dtype, eng, action_pool, k, plan):
    """The dominant kernel of a rollout, measured live: HIP events on the streams its launches are issued on
    (mean over the slices' chains and launches: solo_engine_time_rollout runs the rollout exactly as solo_engine_rollout
    does), same workload (fresh actions every step); one launch = (n / slices) robots x steps_per_launch steps;
    algorithmic bytes per env-step from SURVEY.md 8d."""
    spl, slices = plan['steps_per_launch'], plan['slices']
    kk = min(k, 1000) // spl * spl or k   # (whole launches)
    bufs = eng.rollout_buffers(kk)   # (every step's outputs recorded: the timed region's own call)
    # (a 20-step launch lasts as long as its slowest robot - 0.59 ... 0.98 ms from action sample to action sample: fifteen samples
    # (round 5) put the median ABOVE the wall clock's median of ~700 repeats; now as many samples as 0.15 s of launches hold,
    # 101 for the driver's 20-step launch, fresh actions each)
    samples = KERNEL_MS_SAMPLES if k <= 100 else 5
    ks = sorted(eng.time_rollout(action_pool(kk), abi.STEP_ALL, out=bufs) for _ in range(samples))
    kern_ms = statistics.median(ks)
    env_steps_per_launch = (n // slices) * spl
    bytes_per_launch = BYTES_PER_ENV_STEP[dtype] * env_steps_per_launch
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
    pmc = pmc_profile(dtype, spl, slices)
    traffic = pmc['hbm_bytes_per_env_step'] * env_steps_per_launch if pmc.get('hbm_bytes_per_env_step') else None
    # (waves resident per SIMD: 128 VGPRs allow four in both precisions since round 5)
    note, secondary = secondary_bound(pmc, env_steps_per_launch, slices, kern_ms, min(n / NUM_SIMDS, plan.get('waves_per_simd') or 4), dtype)
    return {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic,
            'traffic_note': pmc.get('traffic_note'),
            'traffic_profile': None if not pmc else {'entry': pmc.get('profile_key'), 'measured_on_this_launch_geometry': pmc.get('geometry_match'),
                                                     'stale': bool(pmc.get('stale')), 'kernel_source_hash': pmc.get('source_hash'),
                                                     'profile_kernel_source_hash': pmc.get('profile_hash') or pmc.get('kernel_source_hash')},
            'kernel': eng.kernel_name, 'kernel_ms': kern_ms, 'kernel_ms_samples': len(ks), 'kernel_ms_min': ks[0], 'kernel_ms_max': ks[-1],
            'kernel_ms_p10_p90': [ks[len(ks) // 10], ks[(9 * len(ks)) // 10]],
            'kernel_ms_note': 'HIP events around the launch chain of every slice, on the stream it is launched on; mean per '
                              'launch over slices and launches (the step kernel is the only kernel of a launch: its epilogue writes the outputs)',
            'bytes_per_env_step': BYTES_PER_ENV_STEP[dtype],
            'algorithmic_bytes_per_launch': bytes_per_launch, 'env_steps_per_launch': env_steps_per_launch,
            'concurrent_launch_chains': slices, 'achieved_all_chains': achieved * slices,
            'launch_plan': plan, 'note': note, 'secondary': secondary}

  def critical_path_floor(dtype, k, plan):
    """The bound that binds at the metric's size is the launch's CRITICAL PATH: with every robot resident a fused launch lasts as
    long as its slowest robot's K sequential steps.  Its floor, measured live: the same workload with ONE
    wave per SIMD (resident_robots / waves_per_simd robots: nobody shares a SIMD, so the launch is its slowest robot's K steps
    at a lone wave's issue rate) - HIP events, median of 51 launches.  kernel_ms / floor_ms says what sharing the SIMDs with
    three other robots costs that path."""
    lone = max(64, (plan.get('resident_robots') or 4096) // max(1, plan.get('waves_per_simd') or 4))
    if lone >= n:
      return None
    tdtype = torch.float32 if dtype == 'float32' else torch.float64
    env1 = build_env(lone, local_rank, dtype, steps_per_launch=args.steps_per_launch, rollout_streams=1, migrate_steps=0)
    e1 = env1.engine
    g1 = torch.Generator(device=dev).manual_seed(rank_seed(4321, rank))
    spl = plan['steps_per_launch']
    desynchronise_episodes(e1, g1, chunk=spl)
    bufs = e1.rollout_buffers(spl)
    pool = lambda: (torch.rand(spl, lone, abi.NUM_JOINTS, device=dev, dtype=tdtype, generator=g1) * 2 - 1) * two_pi
    e1.time_rollout(pool(), abi.STEP_ALL, out=bufs)
    fs = sorted(e1.time_rollout(pool(), abi.STEP_ALL, out=bufs) for _ in range(51 if spl <= 100 else 5))
    env1._close()
    return {'floor_ms': statistics.median(fs), 'floor_ms_min': fs[0], 'floor_ms_max': fs[-1], 'floor_robots': lone, 'floor_samples': len(fs),
            'floor_note': 'the same launch (%d steps, same workload, steady state) with ONE wave per SIMD - %d robots: the slowest robot\'s '
                          'sequential steps with its SIMD to itself; the launch at the metric\'s size cannot be shorter than its own slowest '
                          'robot alone (the slowest of %d robots is slower than the slowest of %d: a lower bound of the bound)' % (spl, lone, n, lone)}

  times, stats, eng, env, action_pool, plan = timed(args.dtype, k, False, args.min_seconds, args.max_repeats)
  log('timed region done: %d repeats, stats all-reduce ok (episodes %.0f)' % (len(times), float(stats[2])))
  elapsed = statistics.median(times)
  roof = roofline(args.dtype, eng, action_pool, k, plan)
  gen_ms = action_generation_ms(action_pool, k)
  eng_cfg_ulp = eng.cfg.solver_ulp_tolerance
  env._close()
  if rank == 0 and not EMU and not args.no_extra:
    floor = critical_path_floor(args.dtype, k, plan)
    if floor:
      roof.update(floor)
      roof['achieved_over_floor'] = floor['floor_ms'] / roof['kernel_ms']   # 1.0 = the launch is as short as its slowest robot alone would make it

  extra = {}
  if not args.no_extra:
    ke = min(k, 500)  # (bounded intervals: f64 is ~2x, single-step launches ~2-4x slower per step)
    other = 'float32' if args.dtype == 'float64' else 'float64'
    tag = {'float32': 'f32', 'float64': 'f64'}
    # the OTHER precision on the same workload and rollout path, with its own roofline block (f64 is the reference's
    # precision - PyBullet computes in double - and the kernel of `value` by default; f32 is the opt-in fast mode)
    to, so_, go, eo, poolo, plano = timed(other, ke, False, 0.3, 10)
    extra['value_' + tag[other]] = world * n * ke / statistics.median(to)
    extra['value_%s_note' % tag[other]] = ('same workload and rollout path in %s (%s), median of %d repeats of %d steps'
                                           % (other, 'the opt-in fast mode; NOT the reference\'s precision' if other == 'float32'
                                              else 'the reference\'s precision, SURVEY.md §8', len(to), ke))
    extra['roofline_' + tag[other]] = roofline(other, go, poolo, ke, plano)
    extra['episodes_' + tag[other]] = summarize(so_.cpu().numpy())
    eo._close()
    for dt in (args.dtype, other):
      sfx = '' if dt == args.dtype else '_' + tag[dt]
      # pybullet's documented default solverResidualThreshold (1e-7 [recalled]) as an OPT-IN: off in `value`
      tr, _, _, er, _, _ = timed(dt, ke, False, 0.3, 10, residual_threshold=1e-7)
      extra['value_residual_1e-7' + sfx] = world * n * ke / statistics.median(tr)
      er._close()
      # ... and with the warm start on top of it (SoloConfig.solver_warm_start = 0.85, [recalled] Bullet's rigid-body factor)
      tw, _, _, ew, _, _ = timed(dt, ke, False, 0.3, 10, residual_threshold=1e-7, warm_start=0.85)
      extra['value_residual_warmstart' + sfx] = world * n * ke / statistics.median(tw)
      ew._close()
      tcl, _, _, ecl, _, _ = timed(dt, ke, True, 0.3, 10)
      extra['value_closed_loop' + sfx] = world * n * ke / statistics.median(tcl)
      ecl._close()
    extra['value_residual_note'] = ('the same rollout with SoloConfig.solver_residual_threshold = 1e-7 (pybullet\'s documented default; the '
                                    'Gauss-Seidel iteration ends after a sweep whose largest squared velocity-level change is below it), in the '
                                    'headline precision and (suffix) the other one; NOT the configuration of `value`: '
                                    'without warm starting it leaves a resting robot jittering at 5e-5 rad/s, where the reference\'s recorded '
                                    'rest state has 1e-11')
    extra['value_residual_warmstart_note'] = ('residual threshold 1e-7 + SoloConfig.solver_warm_start = 0.85: every step starts from 0.85 x the impulses the '
                                              'previous one ended with; the cache costs 2 x 64 reals = %d (f64) / 512 (f32) B of memory traffic per env-step ON TOP of '
                                              'the algorithmic %d / 385 B (SURVEY.md 8d: reported separately); NOT the configuration of `value`, and it does not '
                                              'make the threshold solver rest either (profiles/round4_rest_drift.log)' % (1024, 765))
    extra['value_closed_loop_note'] = ('one solo_engine_step launch per env step (outputs evaluated in that launch), the '
                                       'granularity of Solo8VanillaEnv.step (solo8v2vanilla.py:72-102), actions pre-generated, '
                                       'no host synchronisation between steps; median over repeats of %d steps; headline precision and (suffix) the other one' % ke)
    # round 5's convergence tolerance (2 half-ulps; the f64 default is 512 since round 6: core/configs.py says why), for continuity
    if args.dtype == 'float64':
      tu, _, _, eu, _, _ = timed(args.dtype, ke, False, 1.0, 2000, ulp_tolerance=2)   # (a sample like `value`'s: a launch lasts 0.55 ... 0.8 ms from repeat to repeat)
      extra['value_ulp_tolerance_2'] = world * n * ke / statistics.median(tu)
      extra['value_ulp_tolerance_2_note'] = ('the same rollout with SoloConfig.solver_ulp_tolerance = 2 half-ulps (1.1e-16 relative), the default up to round 5; `value` runs '
                                             'with the f64 default of round 6, 512 half-ulps = 5.7e-14 relative - below the rounding noise between two f64 formulations of one '
                                             'step (1e-13), rest behaviour and parity against the oracle\'s 50 plain sweeps unchanged (profiles/round6_ulp_tolerance_sweep.log)')
      eu._close()
    # rounds 1-2 timed the first steps of 4096 synchronised episodes (every robot freshly reset, nobody terminating):
    # the same kernels under that lighter regime, so that this round's line can be compared with theirs
    tsy, _, _, esy, _, _ = timed(args.dtype, ke, False, 0.3, 10, steady=False)
    extra['value_synchronised_start'] = world * n * ke / statistics.median(tsy)
    extra['value_synchronised_start_note'] = ('the regime rounds 1-2 reported as `value`: all robots at the start of an episode (no steady-state '
                                              'preparation, no episode ends inside the window); NOT the configuration of `value`')
    esy._close()

  if rank == 0:
    value = world * n * k / elapsed
    line = {
      'metric': METRIC, 'value': value, 'unit': 'env-steps/s',
      'n_gpus': world, 'steps': k, 'warmup': w, 'ms_per_step': elapsed / k * 1e3,
      'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
      'dtype': 'f32' if args.dtype == 'float32' else 'f64', 'data': 'synthetic',
      'config': {'workload': 'BASELINE configs[1]: %d Solo8 envs/GPU, flat ground, POSITION_CONTROL, '
                             'U(-2pi,2pi) actions, TorsoIMU+MotorEncoder obs, stand reward, '
                             'TimeBasedTermination(%d)+auto-reset, steady state (episode phases spread uniformly by %d untimed steps), dt=1e-3, <= 50 PGS sweeps (to convergence)' % (n, MAX_STEPS, MAX_STEPS),
                 'envs_per_gpu': n, 'steps_per_launch': plan['steps_per_launch'], 'rollout_streams': plan['slices'], 'migrate_steps': plan['migrate_steps'],
                 'launches_per_slice': plan['launches'], 'waves_per_simd': plan.get('waves_per_simd'),
                 'solver_iterations': 50, 'solver_ulp_tolerance': int(eng_cfg_ulp),
                 'launch_policy': 'chosen by the engine (SoloConfig -1 defaults; solo_engine_plan)' if (args.steps_per_launch, args.rollout_streams, args.migrate_steps) == (-1, -1, -1) else 'command line',
                 'parallelism': 'env-batch sharded x%d, RCCL all-reduce of return stats only' % world},
      'timing': {'repeats': len(times), 'statistic': 'median', 'stats_reduction_inside_timed_region': bool(distributed), 'min_ms_per_step': min(times) / k * 1e3,
                 'max_ms_per_step': max(times) / k * 1e3, 'value_best_repeat': world * n * k / min(times),
                 'value_worst_repeat': world * n * k / max(times), 'first_repeats_ms_per_step': [t / k * 1e3 for t in times[:6]],
                 'action_generation_ms_per_repeat': gen_ms, 'value_including_action_generation': world * n * k / (elapsed + gen_ms * 1e-3),
                 'action_generation_note': 'the K x N x 12 actions of a repeat are drawn on the device (torch generator) in FRONT of the barrier: outside the timed '
                                           'region, as the contract prescribes for inputs (resident in HBM when the region starts); measured on its own, device sync on both sides',
                 'note': 'W warm-up steps and one untimed repeat of the timed call first; then each repeat = exactly K steps '
                         'between barrier + device sync on both sides (max over ranks), fresh actions, the simulation '
                         'continues from repeat to repeat; with more than one rank the statistics all-reduce is inside every repeat and is the '
                         'closing barrier (it completes on a rank only after every rank has contributed), followed by the device sync'},
      'roofline': roof,
      'episodes': summarize(stats.cpu().numpy()),
    }
    line.update(extra)
    line['cpu_baseline'] = None if args.no_cpu_baseline else cpu_baseline(n)
    if line['cpu_baseline'] is not None:
      line['cpu_baseline']['pybullet'] = pybullet_baseline()
    print(json.dumps(line), flush=True)
  if distributed:
    dist.barrier()
    log('final barrier ok')
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
