"""bench.py launches its own ranks for --gpus N > 1 (the driver runs `python bench.py --gpus N`
without torch.distributed.run).  Without a GPU every rank must fail loudly with the engine's own
message - not the launcher - and the parent must report the failure."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
  e = dict(os.environ)
  for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'TORCHELASTIC_RUN_ID'):
    e.pop(k, None)
  e.update(env or {})
  return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=e, capture_output=True, text=True,
                        timeout=600)


def test_self_launch_fails_loudly_without_gpu():
  import torch
  if torch.cuda.is_available():
    import pytest
    pytest.skip('CPU-side check (a GPU box runs the real thing)')
  r = _run(['--gpus', '2', '--steps', '20', '--warmup', '5'])
  assert r.returncode != 0
  # (both ranks fail at once; the launcher stops the rest as soon as it sees the first failure)
  assert 1 <= r.stderr.count('needs an MI355X') <= 2 and 'torch.distributed.run' not in r.stderr
  assert '[bench launcher] rank' in r.stderr
  r = _run(['--gpus', '1', '--steps', '20'])
  assert r.returncode != 0 and 'needs an MI355X' in r.stderr


def test_world_size_mismatch_is_reported():
  r = _run(['--gpus', '2'], env={'RANK': '0', 'WORLD_SIZE': '3'})
  assert r.returncode != 0 and 'WORLD_SIZE=3' in r.stderr


def test_a_dying_peer_rank_stops_rank_zero_at_once():
  """A rank > 0 that exits non-zero (no GPU, import error) must not leave rank 0 - and the parent -
  waiting in a rendezvous until some timeout: the launcher polls every child and stops the rest."""
  import time
  import pytest
  sys.path.insert(0, ROOT)
  import bench
  child = [sys.executable, '-c',
           'import os, sys, time\n'
           'if os.environ["RANK"] == "0":\n'
           '  print("rank0 alive", flush=True); time.sleep(300)\n'
           'else:\n'
           '  time.sleep(1); sys.exit(3)\n']
  t0 = time.time()
  with pytest.raises(SystemExit) as e:
    bench.launch_ranks(2, child_cmd=child)
  assert time.time() - t0 < 60
  assert e.value.code not in (0, None)


def test_two_rank_bench_end_to_end_on_the_emulator():
  """`python bench.py --gpus 2` as the driver starts it - the self-launched ranks, the process group (gloo here, RCCL on
  the node), per-rank seeds, the timed repeats with the statistics all-reduce as their closing barrier, max over
  ranks, the roofline and cpu_baseline blocks, ONE JSON line from rank 0 - on the CPU wave emulator of the product
  kernel source (SOLO_BENCH_ENGINE=emu: a few robots per rank, short episodes).  A launcher regression must not wait
  for an 8-GPU node to show."""
  import json
  r = _run(['--gpus', '2', '--steps', '6', '--warmup', '2', '--envs-per-gpu', '4', '--no-extra', '--max-repeats', '2', '--min-seconds', '0'],
           env={'SOLO_BENCH_ENGINE': 'emu', 'SOLO_BENCH_MAX_STEPS': '5', 'SOLO_CPU_BASELINE_SECONDS': '1', 'SOLO_CPU_BASELINE_THREADS': '2'})
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.strip() and not l.startswith('[Gloo]')]  # (gloo announces itself on stdout)
  assert len(lines) == 1, r.stdout
  d = json.loads(lines[0])
  assert d['n_gpus'] == 2 and d['steps'] == 6 and d['warmup'] == 2 and d['dtype'] == 'f64' and d['scaling'] == 'weak'
  assert d['value'] > 0 and abs(d['value'] - 2 * 4 * 6 / (d['ms_per_step'] * 6e-3)) < 1e-6 * d['value']
  assert d['roofline']['bound'] == 'hbm' and d['roofline']['achieved'] > 0 and d['roofline']['bytes_per_env_step'] == 765
  assert d['cpu_baseline']['kind'] == 'port' and d['cpu_baseline']['value'] > 0
  assert d['episodes']['episodes'] > 0                      # both ranks' episodes went through the all-reduce
  assert d['timing']['stats_reduction_inside_timed_region'] is True
  assert "init_process_group('gloo') ok: world_size 2" in r.stderr and 'final barrier ok' in r.stderr


def test_a_rank_that_dies_after_init_process_group_stops_the_others():
  """the launcher's kill path was exercised only for EARLY failures.  Here both ranks rendezvous and initialise
  the process group (gloo), do one collective - and then rank 1 exits non-zero while rank 0 is inside the next barrier, which
  would otherwise wait for its timeout: the launcher sees the dead rank, stops rank 0 and reports the failing rank's status."""
  import time
  import pytest
  sys.path.insert(0, ROOT)
  import bench
  child = [sys.executable, '-c',
           'import os, sys, time, datetime\n'
           'import torch, torch.distributed as dist\n'
           'r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])\n'
           'dist.init_process_group("gloo", rank=r, world_size=w, timeout=datetime.timedelta(seconds=600))\n'
           't = torch.ones(1); dist.all_reduce(t); assert t.item() == w\n'
           'print("rank %d initialised" % r, flush=True)\n'
           'if r == 1:\n'
           '  time.sleep(1); os._exit(7)\n'
           'dist.barrier()\n'          # rank 0: waits for a peer that is gone
           'time.sleep(300)\n']
  t0 = time.time()
  with pytest.raises(SystemExit) as e:
    bench.launch_ranks(2, child_cmd=child)
  assert time.time() - t0 < 90
  assert e.value.code == 7 or e.value.code not in (0, None)
