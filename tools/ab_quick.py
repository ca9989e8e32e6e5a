"""Same-call A/B of engine-library builds on the bench workload (boxes differ by 2-3 %: only numbers from ONE gpurun
call compare).  Every library is measured in a child process of its own (SOLO_HIP_LIB is read at import), the
libraries alternate, REPS rounds:

  gpurun -- python tools/ab_quick.py [--reps 3] [--dtype float64] [--legs k20,closed,s250] [--n 4096] libA.so libB.so ...

(library names are files under gym_solo_amd/csrc - `@TREE/libX.so`: under TREE/gym_solo_amd/csrc, with TREE's python package:
another commit's sources unpacked under the repo root -; a name may carry settings: libX.so:migrate=0:streams=2:spl=250:n=8192;
what is not given is the engine's choice).  Legs: k20 = the driver's geometry (a rollout of 20 steps), closed = one
solo_engine_step launch per env step (20 steps), s250 = a rollout of 1000 steps (fused launches of 250).
Prints env-steps/s by wall clock (median of the repeats, barrier + device sync on both sides as bench.py does) and the
kernel's duration by HIP events."""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(spec):
  sys.path.insert(0, ROOT)
  import torch
  import bench
  from gym_solo_amd import abi
  dtype, n = spec['dtype'], spec['n']
  tdt = torch.float32 if dtype == 'float32' else torch.float64
  out = {}
  for leg in spec['legs']:
    closed = leg == 'closed'
    k = 1000 if leg == 's250' else 20
    # (migrate= / streams= / spl= override; else the engine chooses - SoloConfig's -1 defaults, Engine.plan(k))
    env = bench.build_env(n, 0, dtype, steps_per_launch=1 if closed else spec.get('spl', -1), rollout_streams=1 if closed else spec.get('streams', -1),
                          migrate_steps=0 if closed else spec.get('migrate', -1))
    eng = env.engine
    gen = torch.Generator(device='cuda').manual_seed(1234)
    bench.desynchronise_episodes(eng, gen)
    pool = lambda steps: (torch.rand(steps, n, abi.NUM_JOINTS, device='cuda', dtype=tdt, generator=gen) * 2 - 1) * 6.283185307179586
    bufs = None if closed else eng.rollout_buffers(k)
    def run(a):
      if closed:
        for i in range(a.shape[0]):
          eng.step(a[i], abi.STEP_ALL)
      else:
        eng.rollout(a, abi.STEP_ALL, out=bufs)
    run(pool(k))
    times = []
    reps = spec['repeats'] if k == 20 else max(3, spec['repeats'] // 6)
    for _ in range(reps):
      a = pool(k)
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      run(a)
      torch.cuda.synchronize()
      times.append(time.perf_counter() - t0)
    rec = {'value': n * k / statistics.median(times), 'best': n * k / min(times)}
    if not closed:
      rec['kernel_ms'] = statistics.median(eng.time_rollout(pool(k), abi.STEP_ALL, out=bufs) for _ in range(5))
      rec['plan'] = eng.plan(k)
    out[leg] = rec
    env._close()
  print('AB_RESULT ' + json.dumps(out), flush=True)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--reps', type=int, default=3)
  ap.add_argument('--repeats', type=int, default=30, help='timed repeats per leg inside a child')
  ap.add_argument('--dtype', default='float64')
  ap.add_argument('--legs', default='k20,closed')
  ap.add_argument('--n', type=int, default=4096)
  ap.add_argument('libs', nargs='+')
  args = ap.parse_args()
  rows = {}
  for rep in range(args.reps):
    for name in args.libs:
      parts = name.split(':')
      spec = {'dtype': args.dtype, 'legs': args.legs.split(','), 'n': args.n, 'repeats': args.repeats}
      for p in parts[1:]:
        key, val = p.split('=')
        spec[key] = val if key == 'dtype' else int(val)
      # (`@TREE/libX.so`: the library AND the python package of another source tree under the repo root - e.g. a `git archive`
      # of an earlier commit with another ABI version; the child is that tree's own copy of this script)
      tree = ROOT
      if parts[0].startswith('@'):
        sub, parts[0] = parts[0][1:].split('/', 1)
        tree = os.path.join(ROOT, sub)
      env = dict(os.environ, SOLO_HIP_LIB=os.path.join(tree, 'gym_solo_amd', 'csrc', parts[0]), SOLO_AB_CHILD=json.dumps(spec))
      try:
        res = subprocess.run([sys.executable, os.path.join(tree, 'tools', 'ab_quick.py')], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
      except subprocess.TimeoutExpired:
        print('%s: TIMEOUT' % name, flush=True)
        raise SystemExit(1)  # (a hung GPU step: no further GPU step in this call)
      line = [l for l in res.stdout.decode().splitlines() if l.startswith('AB_RESULT ')]
      if res.returncode != 0 or not line:
        print('%s: FAILED rc=%d\n%s' % (name, res.returncode, res.stderr.decode()[-2000:]), flush=True)
        raise SystemExit(1)
      out = json.loads(line[0][len('AB_RESULT '):])
      rows.setdefault(name, []).append(out)
      print('%-44s %s' % (name, '   '.join('%s %.4g%s' % (leg, r['value'], (' (kernel %.4f ms; %d x %d steps, %d slice(s), migrate %d)' % (
        r['kernel_ms'], r['plan']['launches'], r['plan']['steps_per_launch'], r['plan']['slices'], r['plan']['migrate_steps'])) if 'kernel_ms' in r else '') for leg, r in out.items())), flush=True)
  print('---- medians over %d rounds (%s, N = %d)' % (args.reps, args.dtype, args.n))
  for name, outs in rows.items():
    print('%-44s %s' % (name, '   '.join('%s %.4g' % (leg, statistics.median(o[leg]['value'] for o in outs)) for leg in outs[0])), flush=True)


if __name__ == '__main__':
  if os.environ.get('SOLO_AB_CHILD'):
    child(json.loads(os.environ['SOLO_AB_CHILD']))
  else:
    main()
