"""GPU suite of the env-level cases (HIP engine through the C-ABI); mirrors
gym_solo/envs/test_solo8v2vanilla.py on the batched API."""
import numpy as np
import pytest

import env_cases as cases

pytestmark = pytest.mark.gpu


def make_env(config=None, **kw):
  import torch
  if not torch.cuda.is_available():
    pytest.fail('GPU tests need a visible MI355X')
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig, Solo8VanillaEnv
  config = config or Solo8VanillaConfig()
  if not getattr(config, '_dtype_pinned', False):
    config.dtype = getattr(make_env, 'dtype', 'float64')
  if not getattr(config, '_num_envs_pinned', False):
    config.num_envs = getattr(make_env, 'num_envs', 64)
  return Solo8VanillaEnv(config=config, **kw)


def test_action_space():
  cases.case_action_space(make_env)


def test_step_no_rewards():
  cases.case_step_no_rewards(make_env)


def test_step_simple_reward():
  cases.case_step_simple_reward(make_env)


def test_action_normalization():
  cases.case_action_normalization(make_env)


def test_action_normalization_float32_bound():
  cases.case_action_normalization_float32_bound(make_env)


def test_reset():
  cases.case_reset(make_env)


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_actions_rest_and_motion(dtype):
  make_env.dtype = dtype
  try:
    cases.case_actions_rest_and_motion(make_env)
  finally:
    make_env.dtype = 'float64'


def test_disjoint_environments():
  cases.case_disjoint_environments(make_env)


@pytest.mark.parametrize('normalize', [False, True])
def test_fused_matches_python_and_oracle_f64(normalize):
  cases.case_fused_matches_python_and_oracle(make_env, steps=40, tol=1e-9,
                                             normalize_observations=normalize)


def test_fused_matches_python_f32():
  """f32 engine: fused obs/reward vs the pull-based python path on the same f32 state."""
  import torch
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig
  cfg = Solo8VanillaConfig()
  cfg.dtype = 'float32'
  cfg._dtype_pinned = True
  env = make_env(config=cfg)
  cases.register_benchmark_workload(env, max_steps=1000)
  rng = np.random.default_rng(5)
  for k in range(50):
    a = torch.as_tensor(rng.uniform(-2 * np.pi, 2 * np.pi, (env.num_envs, 12)))
    o, r, d, _ = env.step(a)
    py_o = env.obs_factory.get_obs_python()
    py_r = env.reward_factory.get_reward_python()
    np.testing.assert_allclose(cases.np_(o), cases.np_(py_o), rtol=0, atol=2e-5)
    np.testing.assert_allclose(cases.np_(r), cases.np_(py_r), rtol=0, atol=2e-5)
    assert not cases.np_(d).any()


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_partial_fused_auto_reset(dtype):
  cases.case_partial_fused_auto_reset(make_env, dtype)


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_reset_restores_motor_targets(dtype):
  cases.case_reset_restores_motor_targets(make_env, dtype)


def test_auto_reset_and_stats():
  """Build extension: in-kernel auto-reset + episodic-return statistics."""
  import torch
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig
  cfg = Solo8VanillaConfig()
  cfg.auto_reset = True
  env = make_env(config=cfg)
  cases.register_benchmark_workload(env, max_steps=9)
  n = env.num_envs
  home = cases.np_(env.engine.snapshot).copy()
  total = np.zeros(n)
  a = torch.zeros(n, 12, dtype=torch.float64)
  for k in range(10):
    o, r, d, _ = env.step(a)
    total += cases.np_(r)
    assert bool(cases.np_(d).all()) == (k == 9)
  env.engine.synchronize()
  st = cases.np_(env.engine.state)
  np.testing.assert_array_equal(st, home)          # restored from the snapshot, counters cleared
  assert (cases.np_(env.engine.term_count) == 0).all()
  stats = cases.np_(env.engine.stats)
  np.testing.assert_allclose(stats[0], total.sum(), rtol=1e-12)
  np.testing.assert_allclose(stats[1], (total ** 2).sum(), rtol=1e-12)
  assert stats[2] == n and stats[3] == 10 * n and stats[5] == 0


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_fused_rollout_equals_single_steps(dtype):
  """solo_engine_rollout_record with steps_per_launch = 7 (fused multi-step launches, state kept
  in LDS) vs one launch per step: same kernel arithmetic -> identical trajectories, outputs
  and auto-reset behaviour."""
  import torch
  from gym_solo_amd import abi
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig
  from gym_solo_amd.workloads import register_benchmark_workload
  tdt = torch.float32 if dtype == 'float32' else torch.float64
  out = {}
  for spl in (1, 7):
    cfg = Solo8VanillaConfig()
    cfg.dtype, cfg._dtype_pinned, cfg.auto_reset, cfg.steps_per_launch = dtype, True, True, spl
    env = make_env(config=cfg)
    register_benchmark_workload(env, max_steps=11)
    env._ensure_program()
    g = torch.Generator(device='cuda').manual_seed(7)
    acts = (torch.rand(30, env.num_envs, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * 6.28
    if spl == 1:
      obs, rew, done = [], [], []
      for k in range(30):
        env.engine.step(acts[k], abi.STEP_ALL)
        obs.append(env.engine.obs.clone()); rew.append(env.engine.reward.clone()); done.append(env.engine.done.clone())
      rec = (torch.stack(obs), torch.stack(rew), torch.stack(done))
    else:
      assert env.engine.steps_per_launch == 7
      rec = env.engine.rollout(acts, abi.STEP_ALL, record=True)
    env.engine.synchronize()
    out[spl] = [t.cpu().numpy() for t in rec] + [env.engine.state.cpu().numpy(), env.engine.term_count.cpu().numpy(),
                                                 env.engine.stats.cpu().numpy()]
    env._close()
  for a, b in zip(out[1], out[7]):
    np.testing.assert_array_equal(a, b)
  assert out[1][2].sum() == 2 * 64  # two episode ends (steps 12 and 24) per robot


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
@pytest.mark.parametrize('spl,streams,k', [(1, 1, 5), (7, 1, 30), (7, 1, 21), (7, 2, 30), (20, 1, 20), (6, 2, 13)])
def test_view_holds_the_last_step_of_a_recorded_rollout(dtype, spl, streams, k):
  """After solo_engine_rollout_record the engine's view (obs / reward / done) holds the LAST step's outputs,
  whatever the launch geometry: the last launch's output epilogue writes them (no copies after the chain),
  a single-step f32 launch evaluates them in place and the rollout copies."""
  import torch
  from gym_solo_amd import abi
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig
  from gym_solo_amd.workloads import register_benchmark_workload
  tdt = torch.float32 if dtype == 'float32' else torch.float64
  cfg = Solo8VanillaConfig()
  cfg.dtype, cfg._dtype_pinned, cfg.auto_reset, cfg.steps_per_launch, cfg.rollout_streams = dtype, True, True, spl, streams
  env = make_env(config=cfg)
  register_benchmark_workload(env, max_steps=k - 1)   # TimeBased(k - 1) fires on step k: the last step ends the first episode
  env._ensure_program()
  eng = env.engine
  g = torch.Generator(device='cuda').manual_seed(11)
  acts = (torch.rand(k, env.num_envs, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * 6.28
  for t in (eng.obs, eng.reward):
    t.fill_(float('nan'))
  eng.done.fill_(7)
  obs, rew, done = eng.rollout(acts, abi.STEP_ALL, record=True)
  eng.synchronize()
  np.testing.assert_array_equal(eng.obs.cpu().numpy(), obs[-1].cpu().numpy())
  np.testing.assert_array_equal(eng.reward.cpu().numpy(), rew[-1].cpu().numpy())
  np.testing.assert_array_equal(eng.done.cpu().numpy(), done[-1].cpu().numpy())
  assert bool(done[-1].all()) and not bool(done[:-1].any())
  env._close()


@pytest.mark.parametrize('dtype,spl', [('float32', 1), ('float32', 7), ('float64', 1), ('float64', 7)])
def test_view_holds_the_last_done_flags_of_a_physics_and_done_only_rollout(dtype, spl):
  """A recording rollout that asks for PHYSICS | DONE only (no observations, no rewards) leaves no step records and
  runs no output epilogue: the step kernel writes the flags into the caller's buffer, and the rollout brings the
  last step's into the engine's view (include/solo_engine.h: "the view holds the last step's outputs")."""
  import torch
  from gym_solo_amd import abi
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig
  from gym_solo_amd.workloads import register_benchmark_workload
  tdt = torch.float32 if dtype == 'float32' else torch.float64
  k = 15
  cfg = Solo8VanillaConfig()
  cfg.dtype, cfg._dtype_pinned, cfg.auto_reset, cfg.steps_per_launch = dtype, True, True, spl
  env = make_env(config=cfg)
  register_benchmark_workload(env, max_steps=k - 1)
  env._ensure_program()
  eng = env.engine
  g = torch.Generator(device='cuda').manual_seed(12)
  acts = (torch.rand(k, env.num_envs, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * 6.28
  eng.done.fill_(7)
  _, _, done = eng.rollout(acts, abi.STEP_PHYSICS | abi.STEP_DONE, record=True)
  eng.synchronize()
  assert bool(done[-1].all()) and not bool(done[:-1].any())
  np.testing.assert_array_equal(eng.done.cpu().numpy(), done[-1].cpu().numpy())
  env._close()



def test_a_fused_rollout_is_capturable_after_reserve():
  """the per-launch scratch of fused launches grows lazily - a device synchronisation and an allocation the FIRST time a
  larger rollout geometry is seen, neither of which a stream capture allows.  solo_engine_reserve (ABI 6) does that work ahead of
  time: after reserve(K) a recording rollout of K steps - several fused launches on two stream slices (fork / join through
  events), migration queues - is captured in a torch.cuda.CUDAGraph WITHOUT ever having run eagerly, and its replays equal an eager
  engine's rollouts bit for bit."""
  import torch
  from gym_solo_amd import abi
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig
  from gym_solo_amd.workloads import register_benchmark_workload
  envs = []
  for _ in range(2):
    cfg = Solo8VanillaConfig()
    cfg.dtype, cfg._dtype_pinned, cfg.auto_reset = 'float64', True, True
    cfg.num_envs, cfg._num_envs_pinned, cfg.steps_per_launch, cfg.rollout_streams, cfg.migrate_steps = 256, True, 16, 2, 5
    env = make_env(config=cfg, copy_outputs=False)
    register_benchmark_workload(env, max_steps=9)
    env._ensure_program()
    envs.append(env)
  eager, captured = envs
  n, k = eager.num_envs, 40
  g = torch.Generator(device='cuda').manual_seed(13)
  static = torch.zeros(k, n, 12, device='cuda', dtype=torch.float64)
  out = captured.engine.rollout_buffers(k)
  captured.engine.reserve(k)                 # (the captured engine has never launched a rollout)
  torch.cuda.synchronize()
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph):
    captured.engine.rollout(static, abi.STEP_ALL, out=out)
  for rep in range(3):
    acts = (torch.rand(k, n, 12, device='cuda', dtype=torch.float64, generator=g) * 2 - 1) * 6.28
    want = eager.engine.rollout(acts, abi.STEP_ALL, record=True)
    static.copy_(acts)
    graph.replay()
    torch.cuda.synchronize()
    for x, y in zip(want, out):
      assert torch.equal(x, y), rep
    assert torch.equal(eager.engine.state, captured.engine.state), rep
  assert captured.engine.stats.cpu().numpy()[6] == 0
  for env in envs:
    env._close()

@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_step_is_capturable_in_a_hip_graph(dtype):
  """The engine enqueues everything on the caller's stream and never synchronises inside step(): an RL library can
  capture Solo8VanillaEnv.step() - next to its policy - in a torch.cuda.CUDAGraph.  Replays of the captured step
  (actions read from a static buffer) equal eager step() calls bit for bit, auto-resets included (TimeBased(6): every
  robot ends an episode inside the 10 steps).  (Measured with tools/gpu_graph_step.py: no time gained - the closed
  loop is bound by the step kernel's slowest robot, not by launches.)"""
  import torch
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig
  from gym_solo_amd.workloads import register_benchmark_workload
  tdt = torch.float32 if dtype == 'float32' else torch.float64
  envs = []
  for _ in range(2):
    cfg = Solo8VanillaConfig()
    cfg.dtype, cfg._dtype_pinned, cfg.auto_reset = dtype, True, True
    env = make_env(config=cfg, copy_outputs=False)
    register_benchmark_workload(env, max_steps=6)
    env._ensure_program()
    envs.append(env)
  eager, captured = envs
  n = eager.num_envs
  g = torch.Generator(device='cuda').manual_seed(3)
  acts = (torch.rand(10, n, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * 6.28
  static = torch.zeros(n, 12, device='cuda', dtype=tdt)
  warm = torch.zeros(n, 12, device='cuda', dtype=tdt)
  for env in envs:   # (first use of the launch outside the capture, the same step on both sides)
    env.step(warm)
  torch.cuda.synchronize()
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph):   # (capture does not execute: the engine's state is untouched)
    obs_g, rew_g, done_g, _ = captured.step(static)
  episodes = 0
  for k in range(10):
    obs_e, rew_e, done_e, _ = eager.step(acts[k])
    static.copy_(acts[k])
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(obs_e, obs_g) and torch.equal(rew_e, rew_g) and torch.equal(done_e, done_g), k
    assert torch.equal(eager.engine.state, captured.engine.state), k
    episodes += int(done_e.sum())
  assert episodes == n
  for env in envs:
    env._close()


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_checkpoint_and_resume_continue_bit_for_bit(dtype):
  """Engine.get_state() / set_state() (checkpoint / resume): a run checkpointed after 30 steps -
  mid-episode, with per-robot parameters and non-trivial counters - and restored into a FRESH engine continues exactly
  like the original: states, observations, rewards, done flags and the episodic statistics of 40 more steps (an
  episode end and its auto-reset among them)."""
  import torch
  from gym_solo_amd import abi
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig
  from gym_solo_amd.workloads import register_benchmark_workload
  tdt = torch.float32 if dtype == 'float32' else torch.float64
  def build():
    cfg = Solo8VanillaConfig()
    cfg.dtype, cfg._dtype_pinned, cfg.auto_reset, cfg.steps_per_launch = dtype, True, True, 10
    env = make_env(config=cfg, copy_outputs=False)
    register_benchmark_workload(env, max_steps=50)
    env._ensure_program()
    return env
  g = torch.Generator(device='cuda').manual_seed(4)
  first, second = build(), build()
  n = first.num_envs
  acts = (torch.rand(70, n, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * 6.28
  mu = torch.rand(n, device='cuda', dtype=tdt, generator=g) * 0.7 + 0.3
  for env in (first, second):   # (both engines get the parameters here; the checkpoint carries them as well)
    env.engine.set_params(abi.PARAM_FRICTION, mu)
  first.engine.rollout(acts[:30], abi.STEP_ALL)
  ck = first.engine.get_state()
  want = first.engine.rollout(acts[30:], abi.STEP_ALL, record=True)
  first.engine.synchronize()
  want_state, want_stats = first.engine.state.clone(), first.engine.stats.clone()
  assert int(want[2].sum()) == n                 # every robot ended an episode in the continued part
  second.engine.set_state(ck)
  got = second.engine.rollout(acts[30:], abi.STEP_ALL, record=True)
  second.engine.synchronize()
  for a, b in zip(want, got):
    assert torch.equal(a, b)
  assert torch.equal(want_state, second.engine.state) and torch.equal(want_stats, second.engine.stats)
  with pytest.raises(ValueError):
    bad = dict(ck); bad['state'] = ck['state'][:-1]
    second.engine.set_state(bad)
  for env in (first, second):
    env._close()


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_checkpoint_carries_the_reset_snapshot(dtype):
  """The reset snapshot reflects the parameters in force at the last settle(), not the current ones (set_params does
  not re-settle): a checkpoint of an engine prepared with set_params(base mass) + settle() restored into a FRESH engine
  (default parameters, default snapshot) must bring the snapshot along, or every auto-reset after the restore puts the
  robots into another pose.  Also: the env's cached outputs are invalidated by the restore."""
  import torch
  from gym_solo_amd import abi
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig
  from gym_solo_amd.workloads import register_benchmark_workload
  tdt = torch.float32 if dtype == 'float32' else torch.float64
  def build():
    cfg = Solo8VanillaConfig()
    cfg.dtype, cfg._dtype_pinned, cfg.auto_reset, cfg.steps_per_launch = dtype, True, True, 10
    env = make_env(config=cfg, copy_outputs=True)
    register_benchmark_workload(env, max_steps=25)
    env._ensure_program()
    return env
  g = torch.Generator(device='cuda').manual_seed(5)
  first, second = build(), build()
  n = first.num_envs
  acts = (torch.rand(60, n, 12, device='cuda', dtype=tdt, generator=g) * 2 - 1) * 6.28
  first.engine.set_params(abi.PARAM_BASE_MASS_SCALE, torch.rand(n, device='cuda', dtype=tdt, generator=g) * 0.4 + 0.8)
  first.engine.settle()
  assert not torch.equal(first.engine.snapshot, second.engine.snapshot)
  first.engine.rollout(acts[:20], abi.STEP_ALL)
  ck = first.engine.get_state()
  want = first.engine.rollout(acts[20:], abi.STEP_ALL, record=True)   # (an episode end + auto-reset for every robot)
  first.engine.synchronize()
  assert int(want[2].sum()) >= n
  stale_obs = second._evaluate_observations().clone()
  second.engine.set_state(ck)
  fresh_obs = second._evaluate_observations().clone()
  assert not torch.equal(stale_obs, fresh_obs)          # the cached observation of the pre-restore state was dropped
  got = second.engine.rollout(acts[20:], abi.STEP_ALL, record=True)
  second.engine.synchronize()
  for a, b in zip(want, got):
    assert torch.equal(a, b)
  assert torch.equal(first.engine.state, second.engine.state) and torch.equal(first.engine.snapshot, second.engine.snapshot)
  with pytest.raises(ValueError):   # a version-2 checkpoint must be whole
    second.engine.set_state({k: v for k, v in ck.items() if k != 'snapshot'})
  assert ck['version'] == second.engine.CHECKPOINT_VERSION == 2
  with pytest.raises(ValueError, match='newer'):
    second.engine.set_state(dict(ck, version=99))
  # a checkpoint of before ABI 4 (no version tag, no snapshot, no warm-start cache) is still accepted: the engine's own
  # snapshot stays in force and the cache starts empty
  old = {k: v for k, v in ck.items() if k not in ('snapshot', 'warm', 'version')}
  keep = second.engine.snapshot.clone()
  second.engine.warm.fill_(1.0)
  second.engine.set_state(old)
  assert torch.equal(second.engine.snapshot, keep) and not second.engine.warm.any() and torch.equal(second.engine.state, ck['state'])
  for env in (first, second):
    env._close()


def test_domain_randomisation_matches_oracle():
  """BASELINE config 4: per-env lateral friction and base-mass scale (changeDynamics per env,
  solo8v2vanilla.py:158-163) — engine.set_params + re-settle vs the oracle with the same params."""
  import torch
  from gym_solo_amd import abi
  from gym_solo_amd.engine import Engine
  from helpers import make_abi
  from oracle import solo_oracle as so
  ca, ma = make_abi('float64')
  n = 32
  rng = np.random.default_rng(4321)
  mu = rng.uniform(0.3, 1.0, n)
  ms = rng.uniform(0.8, 1.2, n)
  eng = Engine(ca, ma, n)
  eng.set_params(abi.PARAM_FRICTION, torch.as_tensor(mu, device='cuda'))
  eng.set_params(abi.PARAM_BASE_MASS_SCALE, torch.as_tensor(ms, device='cuda'))
  eng.settle()
  params = np.zeros((n, 4)); params[:, 0] = mu; params[:, 1] = ms
  np.testing.assert_allclose(eng.params.cpu().numpy(), params)
  ph = so.OraclePhysics(ca, ma)
  st = ph.settle(n, params)
  np.testing.assert_allclose(eng.snapshot.cpu().numpy()[:, :29], st[:, :29], rtol=0, atol=1e-9)
  for k in range(40):
    a = rng.uniform(-2 * np.pi, 2 * np.pi, (n, 12))
    ph.step(st, a, params)
    eng.step(torch.as_tensor(a, device='cuda'), abi.STEP_PHYSICS)
  got = eng.state.cpu().numpy()
  np.testing.assert_allclose(got[:, :29], st[:, :29], rtol=0, atol=1e-9)
  # the randomisation matters: robots with different parameters end up in different states
  assert np.abs(got[0, :29] - got[1, :29]).max() > 1e-4
  eng.close()


def test_client_facade_pull_path_equals_fused_step():
  """The reference's call sequence through the BulletClient-shaped facade
  (setJointMotorControlArray -> stepSimulation -> get_obs / get_reward, solo8v2vanilla.py:87-97)
  gives the same state, observations and rewards as the fused env.step()."""
  import torch
  import gym_solo_amd.client as p
  env_a, env_b = make_env(), make_env()
  for env in (env_a, env_b):
    cases.register_benchmark_workload(env, max_steps=1000)
  rng = np.random.default_rng(11)
  for k in range(15):
    a = torch.as_tensor(rng.uniform(-2 * np.pi, 2 * np.pi, (env_a.num_envs, 12)))
    o, r, d, _ = env_a.step(a)
    env_b.client.setJointMotorControlArray(env_b.robot, np.arange(12), p.POSITION_CONTROL,
                                           targetPositions=a, forces=[env_b.config.motor_torque_limit] * 12)
    env_b.client.stepSimulation()
    ob, _ = env_b.obs_factory.get_obs()
    rb = env_b.reward_factory.get_reward()
    np.testing.assert_array_equal(cases.np_(env_a.engine.state)[:, :29], cases.np_(env_b.engine.state)[:, :29])
    np.testing.assert_array_equal(cases.np_(o), cases.np_(ob))
    np.testing.assert_array_equal(cases.np_(r), cases.np_(rb))
  pos, orn = env_b.client.getBasePositionAndOrientation(env_b.robot)
  assert pos.shape == (env_b.num_envs, 3) and orn.shape == (env_b.num_envs, 4)
  q, qd, _, _ = env_b.client.getJointState(env_b.robot, 2)  # fixed ANKLE joint reads 0
  assert float(q.abs().max()) == 0.0 and float(qd.abs().max()) == 0.0
  assert env_b.client.getNumJoints(env_b.robot) == 12
  assert env_b.client.getJointInfo(env_b.robot, 4)[1] == b'FR_KFE'


@pytest.mark.parametrize('kind', ['incline', 'stairs'])
def test_heightfield_terrain_matches_oracle(kind):
  """BASELINE configs[4]: 10 degree incline / 0.03 m x 0.30 m stairs as a 64x64 heightfield.
  Settle (drop from 0.5 m onto the terrain + fold) and a random rollout, f64 engine vs oracle."""
  import torch
  import helpers
  from gym_solo_amd import abi
  from gym_solo_amd.engine import Engine
  from oracle import solo_oracle as so
  terrain = getattr(helpers, kind + '_terrain')()
  ca, ma = helpers.make_abi('float64')
  n = 16
  eng = Engine(ca, ma, n)
  flat_snapshot = eng.snapshot.cpu().numpy().copy()
  eng.set_terrain(terrain)
  ph = so.OraclePhysics(ca, ma, terrain=terrain)
  st = ph.settle(1)
  snap = eng.snapshot.cpu().numpy()
  np.testing.assert_allclose(snap[:, :29], np.tile(st[:, :29], (n, 1)), rtol=0, atol=1e-8)
  assert np.abs(snap[0, :7] - flat_snapshot[0, :7]).max() > 1e-3  # the ground really changed
  st = np.tile(st, (n, 1))
  rng = np.random.default_rng(9)
  for k in range(40):
    a = rng.uniform(-2 * np.pi, 2 * np.pi, (n, 12))
    ph.step(st, a)
    eng.step(torch.as_tensor(a, device='cuda'), abi.STEP_PHYSICS)
  # (40 steps of contact-rich flailing on a slope amplify last-bit differences ~1e8-fold: the engine's f64
  # rsqrt / sincos are within 2 ulp of the libm results the oracle uses, not bit-identical to them)
  np.testing.assert_allclose(eng.state.cpu().numpy()[:, :29], st[:, :29], rtol=0, atol=3e-8)
  eng.set_terrain(None)
  np.testing.assert_array_equal(eng.snapshot.cpu().numpy(), flat_snapshot)
  eng.close()


def test_all_sixteen_spheres_in_contact_gpu():
  """All 16 collision spheres touching (a robot wedged into a trench narrower than its base, 56
  constraint rows): f64 engine vs oracle over the violent first steps; the solver has no contact cap."""
  import torch
  import helpers
  from gym_solo_amd import abi
  from gym_solo_amd.engine import Engine
  from oracle import solo_oracle as so
  terrain = helpers.trench_terrain()
  ca, ma = helpers.make_abi('float64', settle_steps=0)
  ph = so.OraclePhysics(ca, ma, terrain=terrain)
  st = ph.initial_state(4)
  st[:, abi.S_POS + 2] = 0.08
  st[:, abi.S_Q:abi.S_Q + 8] = [np.pi / 2, np.pi, np.pi / 2, np.pi, -np.pi / 2, -np.pi, -np.pi / 2, -np.pi]
  a = np.tile([np.pi / 2, np.pi, 0, np.pi / 2, np.pi, 0, -np.pi / 2, -np.pi, 0, -np.pi / 2, -np.pi, 0], (4, 1))
  eng = Engine(ca, ma, 4)
  eng.set_terrain(terrain)
  eng.state.copy_(torch.as_tensor(st, device='cuda'))
  dbg = ph.step_debug(st[0].copy(), a[0][[0, 1, 3, 4, 6, 7, 9, 10]])
  assert (dbg.num_rows - 8) // 3 == 16
  for k in range(4):
    ph.step(st, a)
    eng.step(torch.as_tensor(a, device='cuda'), abi.STEP_PHYSICS)
    np.testing.assert_allclose(eng.state.cpu().numpy()[:, :29], st[:, :29], rtol=1e-9, atol=1e-9, err_msg='step %d' % k)
  eng.close()


@pytest.mark.parametrize('n,terrain', [(4096, 'stairs'), (8192, None)])
def test_full_size_properties_f32(n, terrain):
  """BASELINE-size batches — configs[4]: 4096 robots on the stairs heightfield; configs[3]: 8192
  robots on the plane — f32, per-env friction / base-mass randomisation, fused 50-step launches on
  2 stream slices.  Size-independent properties: finite states, unit quaternions, nobody falls
  through the ground, joint rates bounded, reset restores the snapshot, and a replay from the same
  state is bit-identical (determinism, test_solo8v2vanilla.py:141-194)."""
  import torch
  import helpers
  from gym_solo_amd import abi
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig
  cfg = Solo8VanillaConfig()
  cfg.dtype, cfg._dtype_pinned, cfg.steps_per_launch, cfg.rollout_streams = 'float32', True, 50, 2
  if terrain:
    cfg.terrain = getattr(helpers, terrain + '_terrain')()
  make_env.num_envs = n
  try:
    env = make_env(config=cfg)
  finally:
    make_env.num_envs = 64
  cases.register_benchmark_workload(env, max_steps=1000)
  env._ensure_program()
  eng = env.engine
  g = torch.Generator(device='cuda').manual_seed(4321)
  eng.set_params(abi.PARAM_FRICTION, torch.rand(n, device='cuda', generator=g) * 0.7 + 0.3)
  eng.set_params(abi.PARAM_BASE_MASS_SCALE, torch.rand(n, device='cuda', generator=g) * 0.4 + 0.8)
  eng.settle()
  acts = (torch.rand(200, n, 12, device='cuda', generator=g) * 2 - 1) * (2 * np.pi)
  obs, rew, done = eng.rollout(acts, abi.STEP_ALL, record=True)
  st = eng.state.clone()
  assert torch.isfinite(st).all() and torch.isfinite(obs).all() and torch.isfinite(rew).all()
  np.testing.assert_allclose(st[:, 3:7].norm(dim=1).cpu().numpy(), 1.0, atol=1e-5)
  assert float(st[:, 2].min()) > -0.25 and float(st[:, 2].max()) < 3.0
  assert float(st[:, 21:29].abs().max()) < 500 and float((rew < -1e-6).sum()) == 0 and not bool(done.any())
  stats = eng.stats.cpu().numpy()
  assert stats[5] == 0   # nothing diverged
  # determinism: same start state + same actions -> identical bits
  eng.reset()
  eng.rollout(acts[:50], abi.STEP_ALL)
  a = eng.state.clone()
  eng.reset()
  np.testing.assert_array_equal(eng.state.cpu().numpy(), eng.snapshot.cpu().numpy())
  eng.rollout(acts[:50], abi.STEP_ALL)
  assert torch.equal(a, eng.state)
  env._close()


def test_host_termination_with_auto_reset_restarts_the_episode():
  cases.case_host_termination_auto_reset(make_env)
