"""DIAGNOSTIC (CPU): how many constraint rows are LIVE per robot-step in the benchmark workload's steady state?
The f64 step kernel keeps the Delassus columns of the live rows resident in registers; the number of column slots it
reserves decides its VGPR budget and with it its occupancy.  Replays the workload on the f64 C
oracle (flat ground, U(-2pi, 2pi) targets every step, 1000-step episodes with staggered resets) and counts, per
robot-step: 8 motor rows + 3 per sphere closer to the ground than contact_margin + joint-limit rows.

  python tools/live_rows_histogram.py [robots] [steps]  ->  profiles/round4_live_rows_histogram.log"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from gym_solo_amd import abi
from helpers import make_abi
from oracle import solo_oracle as so

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
ca, ma = make_abi('float64')
ph = so.OraclePhysics(ca, ma)
snap = ph.settle(1, threads=1)
st = np.tile(snap, (n, 1))
rng = np.random.default_rng(1234)
phase = rng.integers(0, 1000, n)
radius = np.array([ma.sphere_radius[i] for i in range(ma.num_spheres)])
hist = np.zeros(65, dtype=np.int64)
sph_hist = np.zeros(17, dtype=np.int64)
threads = os.cpu_count() or 1
for k in range(steps):
  ph.step(st, rng.uniform(-2 * np.pi, 2 * np.pi, (n, 12)), threads=threads)
  phase += 1
  done = phase >= 1000
  st[done] = snap
  phase[done] = 0
  if k < 500:   # (let the staggered episodes spread out first)
    continue
  for e in range(n):
    c = ph.sphere_centers(st[e])
    touching = int(((c[:, 2] - radius) < ca.contact_margin).sum())
    q = st[e, abi.S_Q:abi.S_Q + 8]
    limits = int((np.minimum(q + 10.0, 10.0 - q) < ca.joint_limit_margin).sum())
    hist[8 + 3 * touching + limits] += 1
    sph_hist[touching] += 1
tot = hist.sum()
out = ['live constraint rows per robot-step, benchmark workload (flat ground), f64 oracle, %d robots x %d steps (after 500)' % (n, steps - 500),
       'touching spheres: ' + ' '.join('%d:%.2f%%' % (i, 100.0 * sph_hist[i] / tot) for i in range(17) if sph_hist[i])]
cum = 0
for L in range(65):
  if hist[L]:
    cum += hist[L]
    out.append('  L = %2d rows: %6.2f %%   cumulative %7.3f %%' % (L, 100.0 * hist[L] / tot, 100.0 * cum / tot))
print('\n'.join(out))
open(os.path.join(ROOT, 'profiles', 'round4_live_rows_histogram.log'), 'w').write('\n'.join(out) + '\n')
