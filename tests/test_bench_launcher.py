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
  assert r.stderr.count('needs an MI355X') == 2 and 'torch.distributed.run' not in r.stderr
  r = _run(['--gpus', '1', '--steps', '20'])
  assert r.returncode != 0 and 'needs an MI355X' in r.stderr


def test_world_size_mismatch_is_reported():
  r = _run(['--gpus', '2'], env={'RANK': '0', 'WORLD_SIZE': '3'})
  assert r.returncode != 0 and 'WORLD_SIZE=3' in r.stderr
