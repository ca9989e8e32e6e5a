"""MEASUREMENT (CPU, f64 oracle): does pybullet's residual threshold leave a resting robot at rest - cold, and with the
warm start (SoloConfig.solver_warm_start)?  A robot unfolds from the reset pose under zero targets and stands
(z = 0.337); printed: the largest joint rate after 700 / 1500 / 3000 steps for thresholds 1e-7 (pybullet's documented
default) ... 1e-20 (the fixed point) x warm-start factors 0 / 1 / 0.85.   python tools/rest_drift_probe.py > profiles/round4_rest_drift.log"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from helpers import make_abi
from gym_solo_amd import abi
from oracle import solo_oracle as so
base = so.OraclePhysics(*make_abi('float64')).settle(1)
for thr in (1e-7, 1e-10, 1e-14, 1e-20):
  for warm in (0.0, 1.0, 0.85):
    ca, ma = make_abi('float64', solver_residual_threshold=thr, solver_warm_start=warm)
    ph = so.OraclePhysics(ca, ma)
    st = base.copy(); cache = np.zeros((1, 64)); zero = np.zeros((1, 12))
    out = []
    for k in range(3000):
      ph.step(st, zero, warm=cache if warm else None)
      if k in (699, 1499, 2999):
        out.append(np.abs(st[0, abi.S_QD:abi.S_QD + 8]).max())
    print('threshold %g warm %.2f: max |qd| after 700 / 1500 / 3000 steps: %s  z %.4f' % (thr, warm, ['%.1e' % x for x in out], st[0, 2]))
