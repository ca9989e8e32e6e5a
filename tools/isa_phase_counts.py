"""Static instruction counts of the step kernel between its s_memtime phase stamps.

Usage: python tools/isa_phase_counts.py [f|d]   (compiles the -DSOLO_STAMPS assembly itself)
(works on the DIAGNOSTIC -DSOLO_STAMPS assembly; loops are counted once, so the PGS and
contact-build phases show one trip).  Used to see which phases carry the VALU work.
"""
import sys
import os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = '/tmp/solo_stamps.gfx950.s'
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=fast',
                       '-fno-slp-vectorize', *(['-DSOLO_TU_F64', '-mllvm', '-disable-machine-licm'] if (sys.argv[1:] or ['f'])[0] == 'd' else ['-DSOLO_TU_F32']), '-DSOLO_STAMPS', '-S', '--cuda-device-only', '-o', path,
                       os.path.join(ROOT, 'gym_solo_amd/csrc/solo_engine.hip')], stderr=subprocess.DEVNULL)
t = (sys.argv[1:] or ['f'])[0]
lines = open(path).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith('_ZN4solo16solo_step_kernelI%sLb1ELb0ELb0EE' % t)][0]
fe = [i for i, l in enumerate(lines[start:]) if l.startswith('.Lfunc_end')][0] + start
names = ['prologue', 'loads+sync', 'kinematics', 'crba', 'rne bias', 'schur+sum', 'chol+solve', 'rows', 'A build',
         'PGS', 'finish+nan check', 'term+record', 'restart+done', 'loop exit', 'epilogue', 'tail']
new = lambda: {'valu': 0, 'salu': 0, 'lds': 0, 'vmem': 0, 'trans': 0, 'mov': 0, 'rdlane': 0, 'wrlane': 0, 'cnd': 0}
counts, cur = [], new()
for l in lines[start:fe]:
  s = l.strip()
  if not s or s[0] in '.;/' or s.split()[0].endswith(':'):
    continue
  op = s.split()[0]
  if op.startswith('s_memtime'):
    counts.append(cur); cur = new(); continue
  if op.startswith('v_'):
    cur['valu'] += 1
    if op.startswith('v_mov'): cur['mov'] += 1
    if op.startswith('v_readlane') or op.startswith('v_readfirstlane'): cur['rdlane'] += 1
    if op.startswith('v_writelane'): cur['wrlane'] += 1
    if op.startswith('v_cndmask') or op.startswith('v_cmp'): cur['cnd'] += 1
    if any(k in op for k in ('rcp', 'rsq', 'sqrt', 'exp', 'log', 'sin', 'cos', 'div_')):
      cur['trans'] += 1
  elif op.startswith('ds_'): cur['lds'] += 1
  elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): cur['vmem'] += 1
  elif op.startswith('s_'): cur['salu'] += 1
counts.append(cur)
for i, c in enumerate(counts):
  print('%2d %-14s %s' % (i, names[i] if i < len(names) else '?', c))
print('total valu', sum(c['valu'] for c in counts))
