"""BUILD CHECK / REPORT (CPU): scratch accesses (spill stores / reloads) of every step-kernel instantiation of the product
assembly (make -C gym_solo_amd/csrc asm), in total and INSIDE THE STEP LOOP: a reload inside the step sits behind an
s_waitcnt vmcnt(0) that also waits for the step's action load and the previous step's record store.  The step loop is
found through the compiler's own loop annotations of the basic blocks: the loop of depth 1 (non-migrating kernels) or 2
(migrating ones: the task loop is the outer one) that contains the Gauss-Seidel loop (the first s_set_gpr_idx_on).
  python tools/step_body_scratch.py [path/to/solo_engine.gfx950.s]"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def report(path=None):
  text = open(path or os.path.join(ROOT, 'gym_solo_amd', 'csrc', 'solo_engine.gfx950.s')).read()
  out = {}
  for m in re.finditer(r'^(_ZN4solo16solo_step_kernelI\w+?E)E\w*:.*?\n(.*?)^\.Lfunc_end', text, re.S | re.M):
    t, full, resid, migrate = re.search(r'I(\w)Lb(\d)ELb(\d)ELb(\d)E', m.group(1)).groups()
    # basic blocks: label, the loops it belongs to {header: depth}, its instructions
    blocks, cur = [], {'label': None, 'loops': {}, 'lines': []}
    for line in m.group(2).split('\n'):
      lab = re.match(r'^(\.LBB\d+_\d+):(.*)$', line)
      if lab:
        blocks.append(cur)
        cur = {'label': lab.group(1)[2:], 'loops': {}, 'lines': []}
        line = lab.group(2)
      ann = re.search(r';\s+(?:in Loop: Header=|Parent Loop )(BB\d+_\d+) Depth=(\d+)', line)
      if ann and not cur['lines']:
        cur['loops'][ann.group(1)] = int(ann.group(2))
      hdr = re.search(r';\s*=>\s*This (?:Inner )?Loop Header: Depth=(\d+)', line)
      if hdr and not cur['lines']:
        cur['loops'][cur['label']] = int(hdr.group(1))
      if re.match(r'^\s+[a-z]\w+', line) and not line.strip().startswith('.'):
        cur['lines'].append(line.strip())
    blocks.append(cur)
    solver = next(b for b in blocks if any(l.startswith('s_set_gpr_idx_on') for l in b['lines']))
    depth = 2 if migrate == '1' else 1
    step_loop = [h for h, d in solver['loops'].items() if d == depth]
    assert len(step_loop) == 1, (t, full, resid, migrate, solver['loops'])
    inside = [b for b in blocks if step_loop[0] in b['loops']]
    is_scratch = lambda l: l.startswith('scratch_')
    out[(t, int(full), int(resid), int(migrate))] = {
      'scratch_total': sum(is_scratch(l) for b in blocks for l in b['lines']),
      'scratch_in_step_loop': sum(is_scratch(l) for b in inside for l in b['lines']),
      'step_loop_instructions': sum(len(b['lines']) for b in inside), 'step_loop': step_loop[0]}
  return out


if __name__ == '__main__':
  r = report(sys.argv[1] if len(sys.argv) > 1 else None)
  for (t, full, resid, migrate), v in sorted(r.items()):
    print('solo_step_kernel<%s, %s, %s, %s>: scratch accesses %d, inside the step loop (%s, %d instructions) %d' % (
      'float' if t == 'f' else 'double', bool(full), bool(resid), bool(migrate), v['scratch_total'], v['step_loop'], v['step_loop_instructions'], v['scratch_in_step_loop']))
