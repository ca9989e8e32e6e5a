"""DIAGNOSTIC (CPU): where does a step-kernel instantiation spill?  Compiles the device code with line tables and lists
the scratch stores / loads of one kernel by source line.
  python tools/spill_sites.py [mangled-substring, default IdLb1ELb0 = <double, true, false>] [extra hipcc flags...]"""
import os, re, subprocess, sys
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1] if len(sys.argv) > 1 else 'IdLb1ELb0'
out = '/tmp/solo_spill_sites.s'
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=fast', '-fno-slp-vectorize',
                       '-gline-tables-only', '-S', '--cuda-device-only', '-o', out, 'solo_engine.hip'] + sys.argv[2:],
                      cwd=os.path.join(ROOT, 'gym_solo_amd', 'csrc'))
lines = open(out).read().split('\n')
files = {}
for l in lines:
  m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
  if m: files[int(m.group(1))] = m.group(2)
start = [i for i, l in enumerate(lines) if l.startswith('_ZN4solo16solo_step_kernel' + which) and ':' in l][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end')][0]
cur, st, ld = None, Counter(), Counter()
for l in lines[start:end]:
  m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
  if m: cur = '%s:%s' % (files.get(int(m.group(1)), m.group(1)), m.group(2))
  if 'scratch_store' in l: st[cur] += 1
  if 'scratch_load' in l: ld[cur] += 1
print('scratch stores:', sum(st.values()), dict(st))
print('scratch loads :', sum(ld.values()), dict(ld))
