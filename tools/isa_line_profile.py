"""Static VALU/SALU/LDS instruction counts of the f32 (or f64) step kernel per source line.

Usage: python tools/isa_line_profile.py [f|d] [min_count]
Compiles gym_solo_amd/csrc/solo_engine.hip to gfx950 assembly with line tables and attributes
each instruction to the innermost `.loc` in effect (loops are counted once).
"""
import collections, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = (sys.argv[1:] or ['f'])[0]
minc = int((sys.argv[2:] or ['8'])[0])
out = '/tmp/solo_lines.gfx950.s'
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=fast',
                       '-fno-slp-vectorize', *(['-DSOLO_TU_F64', '-mllvm', '-disable-machine-licm'] if (sys.argv[1:] or ['f'])[0] == 'd' else ['-DSOLO_TU_F32']), '-gline-tables-only', '-S', '--cuda-device-only', '-o', out,
                       os.path.join(ROOT, 'gym_solo_amd/csrc/solo_engine.hip')], stderr=subprocess.DEVNULL)
lines = open(out).read().split('\n')
files = {}
for l in lines:
  s = l.strip()
  if s.startswith('.file') and len(s.split()) >= 3 and s.split()[1].isdigit():
    parts = s.split('"')
    files[int(s.split()[1])] = os.path.basename(parts[-2])
start = [i for i, l in enumerate(lines) if l.startswith('_ZN4solo16solo_step_kernelI%sLb1ELb0ELb0EE' % t)][0]
fe = [i for i, l in enumerate(lines[start:]) if l.startswith('.Lfunc_end')][0] + start
acc = collections.defaultdict(lambda: [0, 0, 0])
loc = ('?', 0)
for l in lines[start:fe]:
  s = l.strip()
  if s.startswith('.loc'):
    p = s.split()
    loc = (files.get(int(p[1]), p[1]), int(p[2]))
    continue
  if not s or s[0] in '.;/' or s.split()[0].endswith(':'):
    continue
  op = s.split()[0]
  if op.startswith('v_'): acc[loc][0] += 1
  elif op.startswith('s_'): acc[loc][1] += 1
  elif op.startswith('ds_'): acc[loc][2] += 1
src = {}
tot = [0, 0, 0]
for (f, n), c in sorted(acc.items()):
  for i in range(3): tot[i] += c[i]
  if c[0] < minc: continue
  if f not in src:
    for d in ('gym_solo_amd/csrc', 'include'):
      p = os.path.join(ROOT, d, f)
      if os.path.exists(p): src[f] = open(p).read().split('\n')
  text = src.get(f, [''] * (n + 1))[n - 1].strip()[:110] if f in src else ''
  print('%-22s %4d  valu %4d salu %4d lds %3d  | %s' % (f, n, c[0], c[1], c[2], text))
print('total valu/salu/lds', tot)
