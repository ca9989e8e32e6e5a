"""The generated gfx950 assembly of the product kernels obeys the register-index rule of solo_pgs_gfx950.h: every
s_set_gpr_idx_on / _off is followed by a scalar instruction and sits inside a hand-written loop; the product contains no
compiler-generated indexed sequence (tools/check_gpr_idx.py; hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_register_index_switches_of_every_kernel_instantiation():
  import check_gpr_idx
  subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'gym_solo_amd', 'csrc'), 'asm'], stderr=subprocess.DEVNULL)
  n, errors = check_gpr_idx.check(os.path.join(ROOT, 'gym_solo_amd', 'csrc', 'solo_engine.gfx950.s'))
  assert not errors, '\n'.join(errors)
  assert n >= 100   # (12 step-kernel instantiations x the row updates of their loops)


def test_the_check_sees_violations(tmp_path):
  import check_gpr_idx
  bad = tmp_path / 'bad.s'
  bad.write_text('\n'.join([
    '_Z4kern:', '\t;;#ASMSTART', '\ts_set_gpr_idx_on s4, gpr_idx(SRC0)', '\tv_fma_f32 v1, v64, s5, v1', '\ts_set_gpr_idx_off',
    '\ts_and_b64 s[0:1], s[2:3], s[4:5]', '\t;;#ASMEND',
    '\ts_set_gpr_idx_on s6, gpr_idx(SRC0)', '\ts_nop 0', '\tv_mov_b32_e32 v2, v3', '\ts_set_gpr_idx_off', '\ts_nop 0', '']))
  n, errors = check_gpr_idx.check(str(bad))
  assert n == 4
  assert len(errors) == 3   # a vector instruction behind the switch; two compiler-generated switches outside a loop
