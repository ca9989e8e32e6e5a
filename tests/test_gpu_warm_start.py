"""SoloConfig.solver_warm_start ON THE GPU (an opt-in of the residual-threshold solver: the iteration of a step starts
from the impulses the previous step ended with): the f64 HIP engine against the f64 oracle through the reference-shaped
env - fused steps with observations, rewards, terminations and auto-resets, the cache carried from step to step, emptied
by an episode end, a reset and a restored robot; in stream slices and under robot migration; and the checkpoint."""
import numpy as np
import pytest

from gym_solo_amd import abi

pytestmark = pytest.mark.gpu


def _env(n, dtype, **kw):
  import torch
  if not torch.cuda.is_available():
    pytest.fail('GPU tests need a visible MI355X')
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig, Solo8VanillaEnv
  from gym_solo_amd.workloads import register_benchmark_workload
  cfg = Solo8VanillaConfig()
  cfg.num_envs, cfg.dtype, cfg.auto_reset = n, dtype, True
  cfg.solver_residual_threshold = 1e-7
  max_steps = kw.pop('max_steps', 9)
  for k, v in kw.items():
    setattr(cfg, k, v)
  env = Solo8VanillaEnv(config=cfg, copy_outputs=False)
  register_benchmark_workload(env, max_steps=max_steps)
  env._ensure_program()
  return env


@pytest.mark.parametrize('factor', [1.0, 0.85])
def test_warm_start_matches_oracle_f64(factor):
  import torch
  from helpers import make_abi
  from env_cases import BENCH_REWARD
  from oracle import solo_oracle as so
  n, steps = 48, 33
  env = _env(n, 'float64', solver_warm_start=factor)
  ca, ma = make_abi('float64', auto_reset=True, solver_residual_threshold=1e-7, solver_warm_start=factor)
  oracle = so.OracleEnv(ca, ma, n, [('torso_imu', {}), ('motor_encoder', {})], [(1, BENCH_REWARD)], [('time', 9)], threads=8)
  np.testing.assert_allclose(env.engine.snapshot.cpu().numpy()[:, :29], oracle.snapshot[:, :29], rtol=0, atol=1e-9)
  assert float(env.engine.warm.abs().max()) == 0.0          # the snapshot starts from an empty cache
  rng = np.random.default_rng(5)
  for k in range(steps):
    a = rng.uniform(-2 * np.pi, 2 * np.pi, (n, 12))
    o, r, d, _ = env.step(torch.as_tensor(a, device='cuda'))
    oo, orr, od = oracle.step(a)
    np.testing.assert_allclose(o.cpu().numpy(), oo, rtol=0, atol=1e-9)
    np.testing.assert_allclose(r.cpu().numpy(), orr, rtol=0, atol=1e-9)
    np.testing.assert_array_equal(d.cpu().numpy().astype(bool), od)
    np.testing.assert_allclose(env.engine.warm.cpu().numpy(), oracle.warm, rtol=0, atol=1e-9)
    if od.all():
      assert float(env.engine.warm.abs().max()) == 0.0      # an episode end empties the cache
  assert float(env.engine.warm.abs().max()) > 0
  mask = torch.zeros(n, dtype=torch.uint8, device='cuda'); mask[::2] = 1
  env.engine.reset(mask)
  w = env.engine.warm.cpu().numpy()
  assert np.abs(w[::2]).max() == 0 and np.abs(w[1::2]).max() > 0   # a masked reset empties the cache of the reset robots only
  env._close()


@pytest.mark.parametrize('dtype,spl,streams,migrate', [('float64', 10, 1, 0), ('float64', 20, 1, 5), ('float32', 7, 2, 0), ('float32', 20, 1, 4), ('float32', 7, 1, 0), ('float32', 2, 1, 0), ('float64', 7, 2, 0)])
def test_warm_start_fused_sliced_and_migrating_launches_equal_single_steps(dtype, spl, streams, migrate):
  """The cache travels through fused launches (global memory, step to step), stream slices and robot migration
  (device-coherent accesses): bit-identical to one launch per step."""
  import torch
  tdt = torch.float32 if dtype == 'float32' else torch.float64
  n, k = 512, 40
  a = _env(n, dtype, solver_warm_start=0.85, steps_per_launch=spl, rollout_streams=streams, migrate_steps=migrate, max_steps=13)
  b = _env(n, dtype, solver_warm_start=0.85, steps_per_launch=1, max_steps=13)
  g = torch.Generator(device='cuda').manual_seed(9)
  acts = (torch.rand(k, n, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * 6.2831853
  ra = a.engine.rollout(acts, abi.STEP_ALL, record=True)
  rb = b.engine.rollout(acts, abi.STEP_ALL, record=True)
  torch.cuda.synchronize()
  for x, y in zip(ra, rb):
    assert torch.equal(x, y)
  for name in ('state', 'warm', 'term_count', 'targets'):
    assert torch.equal(getattr(a.engine, name), getattr(b.engine, name)), name
  assert a.engine.stats.cpu().numpy()[6] == 0
  # ... and the checkpoint carries the cache: a fresh engine continues bit for bit
  ck = a.engine.get_state()
  more = (torch.rand(spl, n, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * 6.2831853
  want = a.engine.rollout(more, abi.STEP_ALL, record=True)
  c = _env(n, dtype, solver_warm_start=0.85, steps_per_launch=spl, rollout_streams=streams, migrate_steps=migrate, max_steps=13)
  c.engine.set_state(ck)
  got = c.engine.rollout(more, abi.STEP_ALL, record=True)
  torch.cuda.synchronize()
  for x, y in zip(want, got):
    assert torch.equal(x, y)
  a._close(); b._close(); c._close()
