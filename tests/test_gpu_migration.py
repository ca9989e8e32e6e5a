"""Robot migration inside a fused launch (SoloConfig.migrate_steps; include/solo_engine.h, the queue:
gym_solo_amd/csrc/solo_kernel_params.h) ON THE GPU: chunks of a launch's steps are handed from wave to wave through a
work queue in device memory.  Scheduling only - every result must be bit-identical to the one-robot-per-wave launch,
whatever the batch size, the chunk size, the number of rings (8 = one per XCD, or 1), the stream slices and the
precision; and no wave may ever have given up waiting (slot 6 of the statistics)."""
import numpy as np
import pytest

from gym_solo_amd import abi

pytestmark = pytest.mark.gpu


def _env(n, dtype, spl, streams, migrate, max_steps, **kw):
  import torch
  if not torch.cuda.is_available():
    pytest.fail('GPU tests need a visible MI355X')
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig, Solo8VanillaEnv
  from gym_solo_amd.workloads import register_benchmark_workload
  cfg = Solo8VanillaConfig()
  cfg.num_envs, cfg.dtype, cfg.auto_reset, cfg.steps_per_launch, cfg.rollout_streams = n, dtype, True, spl, streams
  cfg.migrate_steps = migrate
  for k, v in kw.items():
    setattr(cfg, k, v)
  env = Solo8VanillaEnv(config=cfg, copy_outputs=False)
  register_benchmark_workload(env, max_steps=max_steps)
  env._ensure_program()
  return env


@pytest.mark.parametrize('dtype,n,spl,streams,k,chunk', [
  ('float64', 4096, 20, 1, 20, 5),     # the driver's geometry: eight rings, 4096 robots on 3072 wave slots
  ('float32', 4096, 20, 1, 20, 5),
  ('float64', 1000, 20, 1, 40, 3),     # eight rings of 125 robots, ragged last chunk, two launches
  ('float32', 777, 13, 2, 30, 4),      # one ring per slice (not divisible by 8), slices on two streams, ragged everything
  ('float64', 64, 50, 1, 50, 40),      # more steps than one epilogue pass holds (28 in f64); two chunks
  ('float32', 8192, 20, 2, 20, 2),     # many short chunks
])
def test_migration_is_bit_identical(dtype, n, spl, streams, k, chunk):
  import torch
  tdt = torch.float32 if dtype == 'float32' else torch.float64
  ref = _env(n, dtype, spl, streams, 0, 17)
  mig = _env(n, dtype, spl, streams, chunk, 17)
  g = torch.Generator(device='cuda').manual_seed(n + k)
  acts = (torch.rand(k, n, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * 6.2831853
  # spread the episode phases, so that terminations / auto-resets fall into every chunk
  phase = torch.randint(0, 17, (n,), device='cuda', generator=g, dtype=torch.int32)
  for e in (ref, mig):
    e.engine.term_count[:, 0] = phase
  a = ref.engine.rollout(acts, abi.STEP_ALL, record=True)
  b = mig.engine.rollout(acts, abi.STEP_ALL, record=True)
  torch.cuda.synchronize()
  for x, y in zip(a, b):
    assert torch.equal(x, y)
  for name in ('state', 'targets', 'term_count', 'cost', 'obs', 'reward', 'done'):
    assert torch.equal(getattr(ref.engine, name), getattr(mig.engine, name)), name
  sa, sb = ref.engine.stats.cpu().numpy(), mig.engine.stats.cpu().numpy()
  assert sb[6] == 0                                  # no wave gave up waiting for a ring slot
  np.testing.assert_array_equal(sa[[2, 5]], sb[[2, 5]])   # episodes, restored robots: exact counts
  np.testing.assert_allclose(sa[:4], sb[:4], rtol=1e-12)   # (sums of returns: atomic adds in another order)
  assert sa[2] > 0
  ref._close(); mig._close()


def test_migration_physics_only_rollout_and_residual_threshold():
  """The physics-only rollout (client.stepSimulation loops: flags = PHYSICS - on a migrating engine it keeps the
  physics-only, non-migrating instantiation: robots in step with each other, nothing to balance) and pybullet's
  residual threshold (an opt-in with kernel instantiations of its own) under migration."""
  import torch
  n, k = 512, 24
  for resid in (0.0, 1e-7):
    ref = _env(n, 'float64', 24, 1, 0, 1000, solver_residual_threshold=resid)
    mig = _env(n, 'float64', 24, 1, 6, 1000, solver_residual_threshold=resid)
    g = torch.Generator(device='cuda').manual_seed(3)
    acts = (torch.rand(k, n, 12, device='cuda', dtype=torch.float64, generator=g) * 2 - 1) * 6.2831853
    for flags in (abi.STEP_PHYSICS, abi.STEP_ALL):
      ref.engine.rollout(acts, flags)
      mig.engine.rollout(acts, flags)
      torch.cuda.synchronize()
      assert torch.equal(ref.engine.state, mig.engine.state)
      assert torch.equal(ref.engine.cost, mig.engine.cost)
    assert mig.engine.stats.cpu().numpy()[6] == 0
    ref._close(); mig._close()


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_done_flag_of_a_migrating_launch_that_leaves_no_records(dtype):
  """flags = PHYSICS | DONE without observations / rewards, not recording: no step records, no epilogue - the step kernel
  writes the view's done flag itself, ONE byte per robot (done_stride = 0).  A robot's chunks run on waves of different
  XCDs, whose L2s write plain stores back in any order: only the LAST step's flag may be written (round 4 wrote every
  step's: ADVICE r4).  The flag, the counters and the states must equal the one-robot-per-wave launch's - with episode
  ends falling into the first chunks, so that a stale "done" of an early step would show."""
  import torch
  tdt = torch.float32 if dtype == 'float32' else torch.float64
  n, k = 4096, 20
  flags = abi.STEP_PHYSICS | abi.STEP_DONE
  ref = _env(n, dtype, k, 1, 0, 17)
  mig = _env(n, dtype, k, 1, 4, 17)
  g = torch.Generator(device='cuda').manual_seed(11)
  phase = torch.randint(0, 17, (n,), device='cuda', generator=g, dtype=torch.int32)
  for rep in range(3):
    acts = (torch.rand(k, n, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * 6.2831853
    for e in (ref, mig):
      if rep == 0:
        e.engine.term_count[:, 0] = phase
      e.engine.rollout(acts, flags)
    torch.cuda.synchronize()
    assert torch.equal(ref.engine.done, mig.engine.done)
    assert 0 < int(ref.engine.done.sum()) < n
    assert torch.equal(ref.engine.state, mig.engine.state) and torch.equal(ref.engine.term_count, mig.engine.term_count)
  assert mig.engine.stats.cpu().numpy()[6] == 0
  ref._close(); mig._close()


def test_the_engine_chooses_the_launch_geometry():
  """SoloConfig's -1 defaults (round 5: the measured launch policy lives in the engine, not in bench.py): what
  Engine.plan(k) reports for the benchmark's rollouts, that bench.py's default run uses exactly that, and that the
  results are bit-identical to the plainest geometry (one chain of single launches, no migration)."""
  import os, sys
  import torch
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  import bench
  for dtype in ('float64', 'float32'):
    tdt = torch.float32 if dtype == 'float32' else torch.float64
    auto = bench.build_env(4096, 0, dtype, max_steps=17)              # the three knobs left at -1
    assert (auto.engine.cfg.steps_per_launch, auto.engine.cfg.rollout_streams, auto.engine.cfg.migrate_steps) == (-1, -1, -1)
    p20, p1000 = auto.engine.plan(20), auto.engine.plan(1000)
    assert p20['waves_per_simd'] == 4 and p20['resident_robots'] == 4096
    # the driver's run: ONE launch of 20 steps, every robot on a wave slot of its own - no slices, no migration
    assert (p20['steps_per_launch'], p20['launches'], p20['slices'], p20['migrate_steps']) == (20, 1, 1, 0)
    # the default run: launches of 250 steps on two slices
    assert (p1000['steps_per_launch'], p1000['launches'], p1000['slices'], p1000['migrate_steps']) == (250, 4, 2, 0)
    plain = _env(4096, dtype, 60, 1, 0, 17)
    g = torch.Generator(device='cuda').manual_seed(5)
    acts = (torch.rand(60, 4096, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * 6.2831853
    phase = torch.randint(0, 17, (4096,), device='cuda', generator=g, dtype=torch.int32)
    for e in (auto, plain):
      e.engine.term_count[:, 0] = phase
    a = auto.engine.rollout(acts, abi.STEP_ALL, record=True)
    b = plain.engine.rollout(acts, abi.STEP_ALL, record=True)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
      assert torch.equal(x, y)
    assert torch.equal(auto.engine.state, plain.engine.state)
    auto._close(); plain._close()
  # more robots than wave slots: two chunks per launch (one launch), chunks of 25 on one chain (several launches)
  big = bench.build_env(8192, 0, 'float64', max_steps=17)
  p20, p1000 = big.engine.plan(20), big.engine.plan(1000)
  assert (p20['steps_per_launch'], p20['slices'], p20['migrate_steps']) == (20, 1, 10)
  assert (p1000['steps_per_launch'], p1000['launches'], p1000['slices'], p1000['migrate_steps']) == (250, 4, 1, 25)
  big._close()


def test_scratch_and_queues_grow_with_the_rollouts():
  """The record scratch of fused launches and the migration queues are sized for the rollout at hand and grow when a longer
  one comes (round 5): rollouts of 5, 40 and 300 steps on ONE engine - the engine's own geometry, and the same with robot
  migration forced - against an engine that runs one launch per step; bit-identical outputs and states throughout."""
  import torch
  n = 1024
  auto = _env(n, 'float64', -1, -1, -1, 17)
  forced = _env(n, 'float64', -1, 1, 7, 17)
  plain = _env(n, 'float64', 1, 1, 0, 17)
  g = torch.Generator(device='cuda').manual_seed(21)
  phase = torch.randint(0, 17, (n,), device='cuda', generator=g, dtype=torch.int32)
  for e in (auto, forced, plain):
    e.engine.term_count[:, 0] = phase
  for k in (5, 40, 300):
    acts = (torch.rand(k, n, 12, device='cuda', dtype=torch.float64, generator=g) * 2 - 1) * 6.2831853
    want = plain.engine.rollout(acts, abi.STEP_ALL, record=True)
    for e in (auto, forced):
      got = e.engine.rollout(acts, abi.STEP_ALL, record=True)
      torch.cuda.synchronize()
      for x, y in zip(want, got):
        assert torch.equal(x, y), k
      assert torch.equal(plain.engine.state, e.engine.state), k
  assert auto.engine.plan(300)['steps_per_launch'] == 250 and forced.engine.plan(300)['migrate_steps'] == 7
  assert forced.engine.stats.cpu().numpy()[6] == 0
  for e in (auto, forced, plain):
    e._close()


@pytest.mark.parametrize('streams', [1, 2])
def test_ragged_last_launch_needs_more_chunks_than_the_full_ones(streams):
  """the chunk count of a migrating launch is NOT monotone in its step count - a launch of 128 steps with
  migrate_steps = 1 is cut into 64 chunks of 2 (a ring slot has 7 bits for the chunk index), its 65-step ragged tail into 65
  chunks of 1.  Queue regions sized and strided for the full launch's count were overrun by such a tail (beyond 1792 robots:
  past the allocation; with two slices: into the next slice's header).  Now every launch of up to S steps fits its region
  (queue_slots_per_robot): 193 steps at 2048 robots, against the plainest geometry, bit for bit."""
  import torch
  n, k = 2048, 193
  mig = _env(n, 'float64', 128, streams, 1, 17)
  plain = _env(n, 'float64', 1, 1, 0, 17)
  assert mig.engine.plan(k)['migrate_steps'] == 2 and mig.engine.plan(k)['launches'] == 2
  g = torch.Generator(device='cuda').manual_seed(77)
  phase = torch.randint(0, 17, (n,), device='cuda', generator=g, dtype=torch.int32)
  for e in (mig, plain):
    e.engine.term_count[:, 0] = phase
  acts = (torch.rand(k, n, 12, device='cuda', dtype=torch.float64, generator=g) * 2 - 1) * 6.2831853
  want = plain.engine.rollout(acts, abi.STEP_ALL, record=True)
  got = mig.engine.rollout(acts, abi.STEP_ALL, record=True)
  torch.cuda.synchronize()
  for x, y in zip(want, got):
    assert torch.equal(x, y)
  assert torch.equal(plain.engine.state, mig.engine.state) and torch.equal(plain.engine.term_count, mig.engine.term_count)
  assert mig.engine.stats.cpu().numpy()[6] == 0
  mig._close(); plain._close()


def test_reserve_sizes_the_scratch_ahead_of_time():
  """solo_engine_reserve (ABI 6): the lazily grown record scratch / migration queues are sized NOW for rollouts of up to K steps -
  the first long rollout then allocates nothing (the buffers a 300-step rollout uses are the ones reserve() left: same pointers
  are not visible through the ABI, so the observable is that results equal the plain engine's and a second reserve is a no-op)."""
  import torch
  n = 512
  e = _env(n, 'float64', -1, -1, 3, 17)
  plain = _env(n, 'float64', 1, 1, 0, 17)
  e.engine.reserve(300)
  e.engine.reserve(300)
  with pytest.raises(ValueError):
    e.engine.reserve(0)
  g = torch.Generator(device='cuda').manual_seed(5)
  acts = (torch.rand(300, n, 12, device='cuda', dtype=torch.float64, generator=g) * 2 - 1) * 6.2831853
  want = plain.engine.rollout(acts, abi.STEP_ALL, record=True)
  got = e.engine.rollout(acts, abi.STEP_ALL, record=True)
  torch.cuda.synchronize()
  for x, y in zip(want, got):
    assert torch.equal(x, y)
  e._close(); plain._close()
