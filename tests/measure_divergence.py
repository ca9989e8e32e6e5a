"""(measurement script, not collected by pytest; under tests/ because it uses the oracle)
GPU probe: divergence of the f32 engine (and of the f64 engine) from the f64 CPU oracle over 1000
steps in three regimes (not a test; numbers quoted in DESIGN.md)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from gym_solo_amd import abi
from gym_solo_amd.engine import Engine
from helpers import make_abi
from oracle import solo_oracle as so

def actions_for(regime, k, n, rng):
  if regime == 'rest':            # zero targets from the folded rest pose: legs unfold and push
    return np.zeros((n, 12))
  if regime == 'stand-sway':      # smooth stand-up and sway: contact-rich, not chaotic
    t = k * 1e-3
    a = np.zeros((n, 12))
    amp = 0.2 * min(1.0, t / 0.3)
    for leg in range(4):
      s = 1.0 if leg < 2 else -1.0
      a[:, 3 * leg] = s * (0.5 + amp * np.sin(2 * np.pi * 1.0 * t + leg))
      a[:, 3 * leg + 1] = -s * (1.0 + amp * np.sin(2 * np.pi * 1.0 * t + leg))
    return a
  return rng.uniform(-2 * np.pi, 2 * np.pi, (n, 12))   # 'flail': the benchmark's U(-2pi, 2pi)

n = 64
for regime in ('rest', 'stand-sway', 'flail'):
  ca32, ma = make_abi('float32'); ca64, _ = make_abi('float64')
  e32, e64 = Engine(ca32, ma, n), Engine(ca64, ma, n)
  ph = so.OraclePhysics(ca64, ma)
  st = e64.state.cpu().numpy().copy()
  rng = np.random.default_rng(0)
  out = []
  for k in range(1000):
    a = actions_for(regime, k, n, rng)
    ph.step(st, a, threads=16)
    e64.step(torch.as_tensor(a, device='cuda'), abi.STEP_PHYSICS)
    e32.step(torch.as_tensor(a, device='cuda', dtype=torch.float32), abi.STEP_PHYSICS)
    if k + 1 in (1, 10, 100, 300, 1000):
      s64 = e64.state.cpu().numpy(); s32 = e32.state.cpu().numpy().astype(np.float64)
      def rel(x, y, sl):
        d = np.abs(x[:, sl] - y[:, sl]).max(axis=1)
        scale = np.maximum(np.abs(y[:, sl]).max(axis=1), 1.0)
        return np.median(d / scale), (d / scale).max()
      q, qd, base = slice(7, 15), slice(21, 29), slice(0, 7)
      out.append((k + 1, rel(s64, st, q), rel(s32, st, q), rel(s32, st, qd), rel(s32, st, base)))
  print(f'--- {regime}: steps | f64 engine vs oracle q (median,max) | f32 vs oracle: q | qd | base pose')
  for row in out:
    print('  %5d | %.1e %.1e | %.1e %.1e | %.1e %.1e | %.1e %.1e' % (row[0], *row[1], *row[2], *row[3], *row[4]), flush=True)
  print('  final base z median %.3f' % np.median(st[:, 2]))
  e32.close(); e64.close()
