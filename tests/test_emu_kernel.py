"""The PRODUCT kernel source on the CPU fibre emulator (tests/emu): physics parity with the oracle,
fused multi-step launches, and an UndefinedBehaviorSanitizer build (the GPU pool cannot run
sanitizers)."""
import numpy as np
import pytest

from emu_kernel import EmuEngine
from gym_solo_amd import abi
from helpers import make_abi, random_actions
from oracle import solo_oracle as so


def bench_program():
  """The benchmark's SoloProgram, compiled by the host factories from an engine-less env."""
  from test_env_host import make_env
  from gym_solo_amd.workloads import register_benchmark_workload
  env = make_env()
  register_benchmark_workload(env, max_steps=4)
  env._ensure_program()
  return env.engine.program


@pytest.mark.parametrize('dtype,tol', [('float64', 1e-10), ('float32', 2e-3)])
def test_physics_matches_oracle(dtype, tol):
  ca, ma = make_abi(dtype)
  ph = so.OraclePhysics(ca, ma)
  n = 3
  st = np.tile(ph.settle(1), (n, 1))
  e = EmuEngine(ca, ma, n)
  e.state[:] = st
  rng = np.random.default_rng(0)
  for k in range(25):
    a = random_actions(rng, n)
    ph.step(st, a)
    e.step(a, abi.STEP_PHYSICS)
  np.testing.assert_allclose(e.state[:, :29], st[:, :29], rtol=0, atol=tol)


def test_fused_rollout_equals_single_steps():
  """steps_per_launch > 1 is the same arithmetic: bit-identical trajectories and outputs,
  including an auto-reset in the middle of the fused launch."""
  ca, ma = make_abi('float32', auto_reset=True, settle_steps=60)
  prog = bench_program()
  rng = np.random.default_rng(1)
  acts = rng.uniform(-6, 6, (9, 2, 12))
  a = EmuEngine(ca, ma, 2, program=prog)
  a.settle()
  b = EmuEngine(ca, ma, 2, program=prog)
  b.state[:] = a.state
  b.snapshot[:] = a.snapshot
  obs, rew, done = [], [], []
  for k in range(9):
    a.step(acts[k])
    obs.append(a.obs.copy()); rew.append(a.reward.copy()); done.append(a.done.copy())
  fo, fr, fd = b.rollout(acts)
  np.testing.assert_array_equal(np.stack(obs), fo)
  np.testing.assert_array_equal(np.stack(rew), fr)
  np.testing.assert_array_equal(np.stack(done), fd)
  np.testing.assert_array_equal(a.state, b.state)
  np.testing.assert_array_equal(a.term_count, b.term_count)
  np.testing.assert_array_equal(a.stats, b.stats)
  assert fd[4].all() and fd.sum() == 2  # TimeBased(4) fires on the 5th step of each robot


def test_ubsan_build_runs_clean():
  """-fsanitize=undefined,bounds-strict build of the kernel source: aborts on the first report."""
  ca, ma = make_abi('float32', settle_steps=30)
  prog = bench_program()
  e = EmuEngine(ca, ma, 2, program=prog, variant='ubsan')
  e.settle()
  rng = np.random.default_rng(2)
  for k in range(6):
    e.step(rng.uniform(-6, 6, (2, 12)))
  assert np.isfinite(e.state).all() and np.isfinite(e.obs).all()
