"""SOAK + LONG-RUN STATISTICS: the benchmark workload (4096 robots, U(-2 pi, 2 pi) targets every step, TimeBased(1000)
+ in-kernel auto-reset, 250 steps per launch on 2 stream slices) for CHUNKS x 1000 steps in f32 and in f64, the same
action stream through both (f64 receives the f32 numbers).

Two questions: (a) does anything degrade over millions of steps - non-finite states, restores by the divergence guard,
drifting statistics; (b) with ~4e6 episodes per precision the standard error of the mean episodic return is ~8e-5 of
the mean: does the f32 kernel - the throughput headline - simulate the SAME system as the f64 kernel (the parity path,
pinned to the oracle at 1e-9) at that resolution?  Trajectories cannot be compared (the workload is chaotic, DESIGN.md
section 6); distributions can.

usage: python tools/gpu_soak.py [CHUNKS=1000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from gym_solo_amd import abi
from bench import build_env

N = 4096
CHUNKS = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
res = {}
for dtype in ('float32', 'float64'):
  # (the launch geometry is the engine's, as in bench.py; SOAK_MIGRATE=c forces robot migration in chunks of c steps - the queue under load)
  env = build_env(N, 0, dtype, migrate_steps=int(os.environ.get('SOAK_MIGRATE', '-1')), rollout_streams=1 if os.environ.get('SOAK_MIGRATE') else -1)
  eng = env.engine
  g = torch.Generator(device='cuda').manual_seed(20261004)
  t0 = time.perf_counter()
  half = []
  for c in range(CHUNKS):
    a = (torch.rand(1000, N, abi.NUM_JOINTS, device='cuda', dtype=torch.float32, generator=g) * 2 - 1) * (2 * np.pi)
    eng.rollout(a.to(eng.tdtype), abi.STEP_ALL)
    if (c + 1) % 100 == 0 or c + 1 == CHUNKS:
      torch.cuda.synchronize()
      st = eng.stats.cpu().numpy().astype(np.float64)
      finite = bool(torch.isfinite(eng.state[:, :abi.S_RETURN]).all())
      qerr = float((eng.state[:, abi.S_QUAT:abi.S_QUAT + 4].norm(dim=1) - 1).abs().max())
      print('%s: %7d steps, %.1f s: episodes %d, mean return %.5f, std %.4f, mean length %.2f, restored by the divergence guard %d, state finite %s, |quat| - 1 <= %.1e, base z in [%.3f, %.3f]'
            % (dtype, (c + 1) * 1000, time.perf_counter() - t0, st[2], st[0] / max(st[2], 1), np.sqrt(max(0.0, st[1] / max(st[2], 1) - (st[0] / max(st[2], 1)) ** 2)),
               st[3] / max(st[2], 1), st[5], finite, qerr, float(eng.state[:, 2].min()), float(eng.state[:, 2].max())), flush=True)
      half.append(st.copy())
      assert finite and qerr < (1e-5 if dtype == 'float32' else 1e-13)
      assert st[6] == 0   # no wave of a migrating launch ever gave up waiting for a ring slot
      assert st[2] == N * (((c + 1) * 1000) // 1001) and st[3] == 1001.0 * st[2]   # episode accounting: TimeBased(1000) ends an episode at its step 1001
  st = half[-1]
  mean = st[0] / st[2]
  std = np.sqrt(st[1] / st[2] - mean ** 2)
  # first half vs second half of the run: drift?
  mid = half[len(half) // 2 - 1] if len(half) >= 2 else None
  if mid is not None:
    m1 = mid[0] / mid[2]
    m2 = (st[0] - mid[0]) / (st[2] - mid[2])
    print('%s: mean return of the first half %.5f, of the second half %.5f (standard error of each %.5f)' % (dtype, m1, m2, std / np.sqrt(mid[2])), flush=True)
  res[dtype] = (mean, std, st[2], st[5], (time.perf_counter() - t0))
  print('%s: %.3g env-steps in %.1f s (action generation included) = %.3g env-steps/s' % (dtype, N * CHUNKS * 1000.0, res[dtype][4], N * CHUNKS * 1000.0 / res[dtype][4]), flush=True)
  env._close()

(m32, s32, n32, d32, _), (m64, s64, n64, d64, _) = res['float32'], res['float64']
se = np.sqrt(s32 ** 2 / n32 + s64 ** 2 / n64)
print('f32 vs f64: mean return %.5f vs %.5f: difference %.5f = %.2f standard errors (%.1e of the mean); std %.4f vs %.4f (%.2e relative); '
      'robots restored by the divergence guard %d vs %d' % (m32, m64, m32 - m64, (m32 - m64) / se, abs(m32 - m64) / abs(m64), s32, s64, abs(s32 - s64) / s64, d32, d64))
