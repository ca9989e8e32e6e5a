import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from gym_solo_amd import abi
from gym_solo_amd.engine import Engine
from helpers import make_abi
for tol in (0, 2, 512, 4096):
  ca, ma = make_abi('float64', steps_per_launch=100, solver_ulp_tolerance=tol)
  e = Engine(ca, ma, 64)
  zero = torch.zeros(100, 64, 12, device='cuda', dtype=torch.float64)
  out = []
  for k in range(30):
    e.rollout(zero, abi.STEP_PHYSICS)
    if k in (6, 14, 29):
      out.append(float(e.state[:, abi.S_QD:abi.S_QD+8].abs().max()))
  print('solver_ulp_tolerance %d: max |qd| after 700 / 1500 / 3000 zero-target steps: %s, z %.5f' % (tol, ['%.1e' % x for x in out], float(e.state[0, 2])))
  e.close()
