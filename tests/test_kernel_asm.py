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
  # round 6: the region between a switch on and its switch off is straight-line, with the one indexed instruction
  for body, what in ((['s_lshl_b64 s[0:1], -2, s4', 's_waitcnt lgkmcnt(0)', 'v_fma_f32 v1, v64, s5, v1'], 's_waitcnt'),
                     (['s_lshl_b64 s[0:1], -2, s4', 's_cbranch_scc0 .L1', 'v_fma_f32 v1, v64, s5, v1'], 's_cbranch'),
                     (['s_lshl_b64 s[0:1], -2, s4', '.L2:', 'v_fma_f32 v1, v64, s5, v1'], '.L2:'),
                     (['s_lshl_b64 s[0:1], -2, s4', 'v_fma_f32 v1, v64, s5, v1', 'v_mov_b32_e32 v2, v3', 's_nop 0'], '2 vector'),
                     (['s_lshl_b64 s[0:1], -2, s4'], '0 vector')):
    f = tmp_path / 'region.s'
    f.write_text('\n'.join(['_Z4kern:', '\t;;#ASMSTART', '\ts_set_gpr_idx_on s4, gpr_idx(SRC0)'] + ['\t' + b for b in body] +
                            ['\ts_set_gpr_idx_off', '\ts_and_b64 s[0:1], s[2:3], s[4:5]', '\t;;#ASMEND', '']))
    n, errors = check_gpr_idx.check(str(f))
    assert len(errors) == 1 and what in errors[0], (what, errors)
  ok = tmp_path / 'ok.s'
  ok.write_text('\n'.join(['_Z4kern:', '\t;;#ASMSTART', '\ts_set_gpr_idx_on s4, gpr_idx(SRC0)', '\ts_lshl_b64 s[0:1], -2, s4', '\tv_fma_f32 v1, v64, s5, v1',
                           '\ts_set_gpr_idx_off', '\ts_and_b64 s[0:1], s[2:3], s[4:5]', '\t;;#ASMEND', '']))
  assert check_gpr_idx.check(str(ok)) == (2, [])


def test_resource_budget_of_the_product_kernels():
  """What the kernels' occupancy rests on: 128 VGPRs - FOUR waves per SIMD - in both precisions
  (f64 since round 5; round 4: 168 = three), and NO SCRATCH ACCESS INSIDE THE STEP LOOP of any default-solver
  instantiation, migrating ones included: a spill's reload inside the step sits behind an s_waitcnt vmcnt(0) that also
  waits for the step's action load and the previous step's record store (round 4).  What a kernel spills per launch or
  per task (outside the step loop) is bounded.  (tools/step_body_scratch.py finds the step loop through the compiler's
  loop annotations; the residual-threshold kernels - an opt-in - may reload a few values per step.)"""
  import re
  subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'gym_solo_amd', 'csrc'), 'asm'], stderr=subprocess.DEVNULL)
  text = open(os.path.join(ROOT, 'gym_solo_amd', 'csrc', 'solo_engine.gfx950.s')).read()
  found = {}
  # (one metadata block per kernel: "- .agpr_count: ... .name: ... .vgpr_count: N / .vgpr_spill_count: M")
  for m in re.finditer(r'- \.agpr_count:.*?\.name:\s+(\S+)\n.*?\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', text, re.S):
    k = re.search(r'solo_step_kernelI(\w)Lb(\d)ELb(\d)ELb(\d)E', m.group(1))
    if k:
      found[(k.group(1), int(k.group(2)), int(k.group(3)), int(k.group(4)))] = (int(m.group(2)), int(m.group(3)))
  assert len(found) == 12, sorted(found)   # {f, d} x {physics-only, full} x {default solver, residual threshold} + the four migrating ones
  sys.path.insert(0, os.path.join(ROOT, 'tools'))
  import step_body_scratch
  scratch = step_body_scratch.report()
  assert sorted(scratch) == sorted(found)
  for key, (vgprs, spills) in found.items():
    t, full, resid, migrate = key
    assert vgprs <= 128, (key, vgprs)
    assert spills <= 16, (key, spills)
    if not resid:
      assert scratch[key]['scratch_in_step_loop'] == 0, (key, scratch[key])
    else:
      assert scratch[key]['scratch_in_step_loop'] <= 8, (key, scratch[key])
    assert scratch[key]['step_loop_instructions'] > 2000, (key, scratch[key])   # (the loop found IS the step loop)


def test_hand_over_waits_for_its_stores_before_it_publishes():
  """Robot migration: the record / counter stores of a task (device-coherent sc1 stores) must have been acknowledged
  before the ring slot that hands the robot on is written - the order rests on an EXPLICIT s_waitcnt vmcnt(0)
  (solo_wave_ops.h: wave_release_device; a workgroup-scope fence emits no instruction on gfx950).  In the generated
  assembly of every migrating instantiation: the last global store of the kernel (the slot publication) has an
  s_waitcnt vmcnt(0) between it and the sc1 store in front of it."""
  import re
  subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'gym_solo_amd', 'csrc'), 'asm'], stderr=subprocess.DEVNULL)
  text = open(os.path.join(ROOT, 'gym_solo_amd', 'csrc', 'solo_engine.gfx950.s')).read()
  checked = 0
  for m in re.finditer(r'^(_ZN4solo16solo_step_kernelI\w+?Lb1EE)E\w*:.*?\n(.*?)^\.Lfunc_end', text, re.S | re.M):   # kMigrate = true
    lines = [l.strip() for l in m.group(2).split('\n') if re.match(r'^\s+[a-z]', l)]
    stores = [i for i, l in enumerate(lines) if l.startswith(('global_store', 'global_atomic')) and 'sc1' in l]
    assert len(stores) >= 4, m.group(1)
    # the publication: the LAST sc1 store of the kernel (program order: the task loop's end); in front of it, back to the
    # previous sc1 store / atomic, there must be the wait
    last = stores[-1]
    prev = stores[-2]
    between = lines[prev + 1:last]
    assert any(l.startswith('s_waitcnt') and 'vmcnt(0)' in l for l in between), (m.group(1), between[-12:])
    checked += 1
  assert checked == 4   # {f, d} x {default solver, residual threshold}


def test_no_scalar_instruction_in_the_shadow_of_the_row_updates_readlanes():
  """Round 5 (solo_pgs_gfx950.h, tools/microbench/gen_row64_scan.py): a scalar instruction issued behind a vector
  instruction that writes an SGPR waits ~16 cycles for that write - so in every Gauss-Seidel row update of every product
  kernel the broadcast of the row's impulse change (v_readlane ... %[rs]) is followed by the row's independent VECTOR work
  (lam[row] = cand[row], its threshold) before the first scalar instruction: >= 3 vector instructions behind the f64
  row's two readlanes, >= 2 behind the f32 row's one - and the updated row's lane mask is a scalar shift, not a vector compare."""
  import re
  subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'gym_solo_amd', 'csrc'), 'asm'], stderr=subprocess.DEVNULL)
  text = open(os.path.join(ROOT, 'gym_solo_amd', 'csrc', 'solo_engine.gfx950.s')).read()
  rows = {'f': 0, 'd': 0}
  for m in re.finditer(r'^_ZN4solo16solo_step_kernelI(\w)Lb\dELb\dELb\dEE\w*:.*?\n(.*?)^\.Lfunc_end', text, re.S | re.M):
    ins = [l.strip() for l in m.group(2).split('\n') if re.match(r'^\s+[a-z]\w+', l) and not l.strip().startswith('.')]
    for i, l in enumerate(ins):
      if not l.startswith('s_set_gpr_idx_on'):
        continue
      # walk back from the mode switch to the row's s_ff1: the instructions of the row in front of the indexed FMA
      j = i
      while not ins[j].startswith('s_ff1_i32_b64'):
        j -= 1
        assert i - j < 12, ins[i - 12:i + 1]
      head = ins[j:i]
      lanes = [k for k, x in enumerate(head) if x.startswith('v_readlane_b32')]
      assert lanes and len(lanes) == (2 if m.group(1) == 'd' else 1), head
      behind = head[lanes[-1] + 1:]
      vector_run = 0
      for x in behind:
        if not x.startswith('v_'):
          break
        vector_run += 1
      assert vector_run >= (3 if m.group(1) == 'd' else 2), head
      assert not any(x.startswith('v_cmp_eq_u32') for x in head) and any(x.startswith('s_lshl_b64 vcc, 1,') for x in head), head
      rows[m.group(1)] += 1
  assert rows['f'] >= 48 and rows['d'] >= 48, rows   # (six instantiations per precision x the row updates of their walks)
