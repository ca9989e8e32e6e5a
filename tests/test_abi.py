"""The C-ABI library loads on a CPU-only machine and exports every symbol that
include/solo_engine.h declares (no compute calls without a GPU)."""
import ctypes as C
import os
import re

import pytest

from gym_solo_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'solo_engine.h')
LIB = os.path.join(ROOT, 'gym_solo_amd', 'csrc', 'libsolo_hip.so')


def declared_functions():
  text = open(HEADER).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(solo_[a-z_]+)\s*\(', text)))


def test_header_and_ctypes_mirror_agree():
  assert declared_functions() == sorted(abi.ENTRY_POINTS)


def test_struct_sizes_match_header():
  """Compile a tiny C program against the header and compare sizeof() with the ctypes mirror."""
  import subprocess, tempfile
  src = '#include <stdio.h>\n#include "solo_engine.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",' \
        'sizeof(SoloModel),sizeof(SoloConfig),sizeof(SoloObsElem),sizeof(SoloRewardInstr),' \
        'sizeof(SoloProgram),sizeof(SoloStateView));return 0;}\n'
  with tempfile.TemporaryDirectory() as d:
    open(os.path.join(d, 't.c'), 'w').write(src)
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), '-o', os.path.join(d, 't'),
                           os.path.join(d, 't.c')])
    got = [int(x) for x in subprocess.check_output([os.path.join(d, 't')]).split()]
  want = [C.sizeof(t) for t in (abi.SoloModel, abi.SoloConfig, abi.SoloObsElem,
                                abi.SoloRewardInstr, abi.SoloProgram, abi.SoloStateView)]
  assert got == want


def test_field_offsets_match_header():
  """sizeof() alone lets two equal-sized fields swap places unnoticed: compare offsetof() of EVERY field of every
  struct of the boundary (names taken from the ctypes mirror, so a field missing in the header fails the compile)."""
  import subprocess, tempfile
  structs = [t for t in (abi.SoloModel, abi.SoloConfig, abi.SoloObsElem, abi.SoloRewardInstr, abi.SoloProgram,
                         abi.SoloStateView, abi.SoloTerrain, abi.SoloLaunchPlan)]
  lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "solo_engine.h"', 'int main(){']
  want = []
  for t in structs:
    for name, *_ in t._fields_:
      lines.append('printf("%%zu\\n", offsetof(%s, %s));' % (t.__name__, name))
      want.append((t.__name__, name, getattr(t, name).offset))
  lines.append('return 0;}')
  with tempfile.TemporaryDirectory() as d:
    open(os.path.join(d, 't.c'), 'w').write('\n'.join(lines) + '\n')
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), '-o', os.path.join(d, 't'), os.path.join(d, 't.c')])
    got = [int(x) for x in subprocess.check_output([os.path.join(d, 't')]).split()]
  assert len(got) == len(want) > 60
  for (sname, fname, off), g in zip(want, got):
    assert off == g, '%s.%s: ctypes offset %d, header offset %d' % (sname, fname, off, g)


def test_counter_profiles_are_pinned_to_the_kernel_sources(tmp_path, monkeypatch):
  """bench.py quotes HBM traffic / instruction counts from profiles/pmc_traffic.json only when the profile was taken
  on the kernel sources of THIS tree (gym_solo_amd/build_info.py); otherwise the roofline block says stale and carries
  no traffic figure."""
  import json, sys
  sys.path.insert(0, ROOT)
  import bench
  from gym_solo_amd import build_info
  now = build_info.kernel_source_hash()
  assert len(now) == 16 and now == build_info.kernel_source_hash()
  table = {'float64_k20': {'kernel_source_hash': now, 'steps_per_launch': 20, 'launch_chains': 1, 'hbm_bytes_per_env_step': 700.0},
           'float32_k20': {'kernel_source_hash': 'deadbeefdeadbeef', 'steps_per_launch': 20, 'launch_chains': 1, 'hbm_bytes_per_env_step': 330.0}}
  (tmp_path / 'profiles').mkdir()
  (tmp_path / 'profiles' / 'pmc_traffic.json').write_text(json.dumps(table))
  monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
  fresh = bench.pmc_profile('float64', 20, 1)
  assert fresh['stale'] is False and fresh['hbm_bytes_per_env_step'] == 700.0 and fresh['geometry_match'] is True
  stale = bench.pmc_profile('float32', 20, 1)
  assert stale['stale'] is True and 'hbm_bytes_per_env_step' not in stale and stale['profile_hash'] == 'deadbeefdeadbeef'
  # a changed kernel source changes the hash
  monkeypatch.setattr(build_info, 'kernel_source_files', lambda: [HEADER])
  assert build_info.kernel_source_hash() != now


def test_library_exports_every_declared_symbol():
  if not os.path.exists(LIB):
    import subprocess
    subprocess.check_call(['make', '-s', '-C', os.path.dirname(LIB)])
  lib = C.CDLL(LIB)
  for name in declared_functions():
    assert hasattr(lib, name), name
  abi.bind(lib)
  assert lib.solo_abi_version() == abi.ABI_VERSION


def test_product_fails_loudly_without_gpu():
  import torch
  if torch.cuda.is_available():
    pytest.skip('a GPU is visible')
  from gym_solo_amd.engine import Engine, EngineError
  from helpers import make_abi
  ca, ma = make_abi('float32')
  with pytest.raises(EngineError, match='no CPU fallback'):
    Engine(ca, ma, 4)


def test_create_rejects_bad_arguments():
  from gym_solo_amd.engine import load_library
  from helpers import make_abi
  lib = load_library()
  ca, ma = make_abi('float32')
  h = C.c_void_p()
  assert lib.solo_engine_create(C.byref(ca), C.byref(ma), 0, 0, C.byref(h)) == abi.ERR_INVALID_ARG
  ca.restitution = -0.5   # ([0, 1] is accepted - and has no effect: the ground's restitution is 0, include/solo_engine.h)
  assert lib.solo_engine_create(C.byref(ca), C.byref(ma), 4, 0, C.byref(h)) == abi.ERR_INVALID_ARG
  assert b'restitution' in lib.solo_last_create_error()
  # the launch knobs: -1 = the engine chooses, anything below is rejected; the padding word has a message of its own
  for knob in ('steps_per_launch', 'rollout_streams', 'migrate_steps'):
    ca, ma = make_abi('float32')
    setattr(ca, knob, -2)
    assert lib.solo_engine_create(C.byref(ca), C.byref(ma), 4, 0, C.byref(h)) == abi.ERR_INVALID_ARG
    assert knob.encode() in lib.solo_last_create_error()
  ca, ma = make_abi('float32')
  assert (ca.steps_per_launch, ca.rollout_streams, ca.migrate_steps) == (abi.AUTO, abi.AUTO, abi.AUTO)   # the host defaults
  ca.reserved0 = 7
  assert lib.solo_engine_create(C.byref(ca), C.byref(ma), 4, 0, C.byref(h)) == abi.ERR_INVALID_ARG
  assert b'reserved0' in lib.solo_last_create_error()
  ca, ma = make_abi('float32')
  ma.joint_axis[3][1] = 0.0
  ma.joint_axis[3][2] = 1.0
  assert lib.solo_engine_create(C.byref(ca), C.byref(ma), 4, 0, C.byref(h)) == abi.ERR_UNSUPPORTED_MODEL
