"""Long-horizon parity AT BENCHMARK SCALE: the f64 HIP engine against the f64 C oracle over 1000 steps on the batch
sizes and grounds of BASELINE.json's configurations - configs[1] (4096 robots, flat), configs[3] (8192 robots, per-robot
friction ~U(0.3, 1.0) and base-mass scale ~U(0.8, 1.2)), configs[4] (4096 robots on the 10-degree incline and on the
0.03 m x 0.30 m stairs) - in three regimes (rest: zero targets, the robot unfolds and stands; glide: stand-and-sway on
slippery ground - the moving regime in which nine robots in ten are regular; sway: stand-and-sway about a crouch on the
reference's friction, the stress case - each with per-robot amplitude, frequency and phase, so that no two robots do the
same), and over 60 steps of the benchmark's own random actions at the benchmark's batch sizes.  north_star asks for <= 1e-4
relative joint-state divergence over 1000 steps; the f64 engine is held to 1e-8 on q, q-dot and the base pose.
(`rest` on the STAIRS heightfield is left out on purpose: a foot that comes to rest on the edge between a tread cell and
the 31-degree riser cell of the bilinear heightfield sits on a discontinuity of the ground normal - a sliding-mode
equilibrium in which the cell is chosen by the last bit of the position; engine and oracle then differ by 2e-6 after
1000 steps, measured, while a 1e-10 perturbation of the oracle itself stays below 1e-5.)
(The benchmark's own U(-2pi, 2pi) flailing is chaotic - any two arithmetics decorrelate within a few hundred steps -
and is compared as distributions: tests/test_gpu_parity_f32.py.  The f32 engine's one-step joint-RATE error is 1.3e-4
p99 / 3.5e-4 max, i.e. ABOVE 1e-4: profiles/round3_parity_f32_vs_oracle.log - f64 is the parity path.)"""
import numpy as np
import pytest

from gym_solo_amd import abi
from helpers import make_abi, incline_terrain, stairs_terrain

pytestmark = pytest.mark.gpu


def _actions(regime, k0, k1, n, rng_params):
  amp_e, f_e, ph_e = rng_params
  a = np.zeros((k1 - k0, n, 12))
  if regime == 'rest':
    return a
  t = (np.arange(k0, k1) * 1e-3)[:, None]
  ramp = np.minimum(1.0, t / 0.3)
  for leg in range(4):
    s = 1.0 if leg < 2 else -1.0
    w = amp_e[None, :] * ramp * np.sin(2 * np.pi * f_e[None, :] * t + leg + ph_e[None, :])
    if regime == 'glide':   # about the straight-legged stand (what `rest` settles into): hips w, knees -2 w
      a[:, :, 3 * leg] = s * w
      a[:, :, 3 * leg + 1] = -s * 2.0 * w
    else:                   # `sway`: about a crouch
      a[:, :, 3 * leg] = s * (0.5 + w)
      a[:, :, 3 * leg + 1] = -s * (1.0 + w)
  return a


# Three regimes.  `rest`: zero targets, the robot unfolds and stands.  `sway` (the STRESS case): stand-and-sway about a
# crouch on the reference's friction (0.5) - feet that stick and slip in turns: 40 ... 60 % of the robots amplify a 1e-10
# perturbation beyond 1e-7 in ANY arithmetic.  `glide` (round 5): the same sway about the straight-legged stand on
# SLIPPERY ground (friction 0.1; randomised: ~U(0.05, 0.15)) - feet that slide all the time, no stick-slip transitions:
# the moving regime in which (almost) every robot is regular and is held to the hard bar.  (Measured on the oracle,
# profiles/round6_parity_scale_f64.log: smaller sway amplitudes or MORE friction make the motion less regular, not
# more - friction 1.0 leaves 5 % of the robots regular.)
GLIDE_FRICTION = 0.1
# share of the robots that must be regular (the oracle's own 1e-10 twin within 1e-7), per ground and regime
MIN_REGULAR = {'rest': 0.9, 'glide': 0.9, 'sway': 0.35}


@pytest.mark.parametrize('name,n,regime', [
  ('flat', 4096, 'rest'), ('flat', 4096, 'glide'), ('flat', 4096, 'sway'),
  ('randomised', 8192, 'rest'), ('randomised', 8192, 'glide'),   # (`sway` on friction 0.3 slides in bursts: measured once, a third of the robots regular - not a parity statement)
  ('incline', 4096, 'rest'), ('incline', 4096, 'glide'), ('incline', 4096, 'sway'),   # (`rest` on the incline: the reference's friction 0.5 holds on 10 degrees)
  ('stairs', 4096, 'glide'), ('stairs', 4096, 'sway'),
])
def test_thousand_steps_at_benchmark_scale_f64(name, n, regime):
  import torch
  if not torch.cuda.is_available():
    pytest.fail('GPU tests need a visible MI355X')
  from gym_solo_amd.engine import Engine
  from oracle import solo_oracle as so
  import bench
  threads = bench.host_cores()
  kw = {'lateral_friction': GLIDE_FRICTION} if regime == 'glide' else {}
  ca, ma = make_abi('float64', steps_per_launch=50, **kw)
  terrain = {'incline': incline_terrain, 'stairs': stairs_terrain}.get(name, lambda: None)()
  eng = Engine(ca, ma, n)
  rng = np.random.default_rng(4321)
  params = np.zeros((n, 4)); params[:, 0] = ca.lateral_friction; params[:, 1] = 1.0
  if name == 'randomised':
    params[:, 0] = rng.uniform(0.05, 0.15, n) if regime == 'glide' else rng.uniform(0.3, 1.0, n)
    params[:, 1] = rng.uniform(0.8, 1.2, n)
    eng.set_params(abi.PARAM_FRICTION, torch.as_tensor(params[:, 0], device='cuda'))
    eng.set_params(abi.PARAM_BASE_MASS_SCALE, torch.as_tensor(params[:, 1], device='cuda'))
    eng.settle()
  if terrain is not None:
    eng.set_terrain(terrain)   # (re-settles on the new ground)
  if name == 'incline':
    # spread the robots over the slope (the plane z = tan(10 deg) x: a shifted robot is lifted with it)
    dx, dy = rng.uniform(-0.6, 0.6, n), rng.uniform(-0.6, 0.6, n)
    eng.state[:, 0] += torch.as_tensor(dx, device='cuda')
    eng.state[:, 1] += torch.as_tensor(dy, device='cuda')
    eng.state[:, 2] += torch.as_tensor(np.tan(np.radians(10.0)) * dx, device='cuda')
  elif name == 'stairs':
    eng.state[:, 1] += torch.as_tensor(rng.uniform(-0.6, 0.6, n), device='cuda')   # (along the steps)
  ph = so.OraclePhysics(ca, ma, terrain=terrain)
  st = eng.state.cpu().numpy().copy()
  # WHICH robots are in a non-chaotic regime is decided by the oracle alone: every robot is simulated twice, the second
  # copy from a state perturbed by 1e-10 rad in one joint.  A robot whose two copies end within 1e-5 of each other
  # (amplification < 1e5: rounding differences of 1e-16 per step then stay far below 1e-8) is REGULAR and must meet
  # the bar; the others - per-robot sway parameters that tip into a stick-slip gait - amplify round-off 1e8-fold and
  # more in ANY arithmetic, and say nothing about parity.
  twin = np.concatenate([st, st.copy()])
  twin[n:, abi.S_Q + 1] += 1e-10
  params2 = np.concatenate([params, params])
  sway = (rng.uniform(0.10, 0.20, n), rng.uniform(0.6, 1.0, n), rng.uniform(0.0, 2 * np.pi, n))
  for k0 in range(0, 1000, 100):
    a = _actions(regime, k0, k0 + 100, n, sway)
    eng.rollout(torch.as_tensor(a, device='cuda'), abi.STEP_PHYSICS)
    a2 = np.concatenate([a, a], axis=1)
    for k in range(100):
      ph.step(twin, a2[k], params2, threads=threads)
  got, st, st_b = eng.state.cpu().numpy(), twin[:n], twin[n:]
  # the ONE-STEP disagreement of the two formulations (engine: factored star-tree dynamics and whitened rows; oracle: CRBA +
  # dense Cholesky + velocity-space Gauss-Seidel), measured here: one more step of both from the oracle's final states
  eng.state.copy_(torch.as_tensor(st, device='cuda'))
  a_last = _actions(regime, 1000, 1001, n, sway)
  eng.rollout(torch.as_tensor(a_last, device='cuda'), abi.STEP_PHYSICS)
  one = st.copy()
  ph.step(one, a_last[0], params, threads=threads)
  one_gpu = eng.state.cpu().numpy()

  def rel(x, y, sl):
    return np.abs(x[:, sl] - y[:, sl]).max(axis=1) / np.maximum(np.abs(y[:, sl]).max(axis=1), 1.0)
  parts = {'q': slice(7, 15), 'qd': slice(21, 29), 'base pose': slice(0, 7), 'base velocity': slice(15, 21)}
  sens = np.max([rel(st_b, st, sl) for sl in parts.values()], axis=0)     # the oracle's own sensitivity, per robot
  regular = sens < 1e-7
  errs = {k: rel(got, st, sl) for k, sl in parts.items()}
  worst = np.max(list(errs.values()), axis=0)
  ratio = worst / np.maximum(sens, 1e-13)
  eps = np.max([rel(one_gpu, one, sl) for sl in parts.values()], axis=0)   # per robot, one step
  print('   one-step disagreement engine vs oracle: median %.1e p99 %.1e max %.1e; regular robots: error / twin divergence max %.2g p99.9 %.2g; '
        'quiet robots (twin divergence < 1e-9): %d, their error max %.1e' % (np.median(eps), np.quantile(eps, 0.99), eps.max(), ratio[regular].max(),
        np.quantile(ratio[regular], 0.999), int((sens < 1e-9).sum()), worst[sens < 1e-9].max() if (sens < 1e-9).any() else 0.0))
  print('%s n=%d %s: regular robots %.1f %%; engine vs oracle on them: %s; all robots: error / (oracle vs its 1e-10-perturbed twin) max %.2g, '
        'p99 %.2g; chaotic robots: error max %.1e, twin divergence median %.1e' % (
    name, n, regime, 100.0 * regular.mean(), {k: '%.1e' % v[regular].max() for k, v in errs.items()}, ratio.max(), np.quantile(ratio, 0.99),
    worst[~regular].max() if (~regular).any() else 0.0, np.median(sens[~regular]) if (~regular).any() else 0.0))
  assert np.isfinite(got[:, :29]).all()
  # (the stairs: a foot that crosses the edge between a tread cell and a riser cell of the bilinear heightfield crosses a
  # discontinuity of the ground normal - no regime in which the robots MOVE over them keeps nine in ten regular)
  assert regular.mean() > (0.5 if (name, regime) == ('stairs', 'glide') else MIN_REGULAR[regime]), regular.mean()
  # THE BAR, derived per robot from the oracle's own sensitivity (round 6; round 5 had fitted "every regular robot < 3e-8" to two
  # measured outliers).  The twin started 1e-10 away and ended `sens` away: the robot amplifies a perturbation made at step 0 by
  # sens / 1e-10, and one made later by no more.  Engine and oracle - two formulations of the step - disagree by eps per step
  # (measured above on THIS batch: median 1e-13, 99th percentile up to 4e-12), so after 1000 steps a robot's error is at most
  #     K x sens,   K = 1000 steps x eps / 1e-10
  # with eps the 99th percentile of the measured one-step disagreement.  Where that product is small the bar is the hard one:
  #  * every QUIET robot (twin divergence < 1e-9: no amplification to speak of) within 1e-8 on q, q-dot, base pose, base velocity;
  #  * every regular robot (twin divergence < 1e-7) within max(1e-8, K x its own twin divergence);
  #  * and 999 in 1000 of the regular robots within 1e-8 whatever their sensitivity (a statement about the batch, not a bound).
  quiet = sens < 1e-9
  K = max(1.0, 1000.0 * float(np.quantile(eps, 0.99)) / 1e-10)
  bound = np.maximum(1e-8, K * sens)
  print('   bar: K = %.1f; quiet robots %d, worst %.1e (< 1e-8); regular robots: worst error / own bound %.2g' % (
    K, int(quiet.sum()), worst[quiet].max() if quiet.any() else 0.0, (worst[regular] / bound[regular]).max()))
  assert not quiet.any() or worst[quiet].max() < 1e-8, float(worst[quiet].max())
  assert (worst[regular] <= bound[regular]).all(), float((worst[regular] / bound[regular]).max())
  assert np.quantile(worst[regular], 0.999) < 1e-8, {k: float(v[regular].max()) for k, v in errs.items()}
  # ... and for the robots at large, chaotic or not: the engine is closer to the oracle than the oracle is to its own twin
  # that started 1e-10 rad away (rounding differences are ~1e-16 per step: the engine behaves like a perturbation far
  # below 1e-10).  99 % of them, not all: a contact that switches between sticking and sliding (or between two cells of
  # a heightfield) is a discontinuity, which the last bit can trigger in one copy and the 1e-10 perturbation not.
  assert np.quantile(ratio, 0.99) < 1.0 and ratio.max() < 1e4
  # the robots are really doing something, and not all the same thing
  assert np.median(st[:, 2]) > 0.1 and (regime == 'rest' or np.std(st[:, abi.S_Q]) > 1e-3)
  eng.close()


@pytest.mark.parametrize('name,n', [('flat', 4096), ('randomised', 8192)])
def test_benchmark_actions_short_horizon_at_benchmark_scale_f64(name, n):
  """The benchmark's OWN workload at the benchmark's batch sizes (BASELINE configs[1] and [3]): U(-2 pi, 2 pi) targets
  every step - contact-rich flailing, chaotic: round-off grows ~e^(50 t) - over the 60 steps in which f64 round-off
  stays below 1e-9 (tests/test_gpu_physics.py::test_random_rollout_matches_oracle_f64 is the 64-robot version), as ONE
  fused launch: f64 HIP engine vs the f64 oracle on q, q-dot, base pose and base velocity of EVERY robot."""
  import torch
  if not torch.cuda.is_available():
    pytest.fail('GPU tests need a visible MI355X')
  from gym_solo_amd.engine import Engine
  from oracle import solo_oracle as so
  import bench
  from helpers import random_actions
  threads = bench.host_cores()
  ca, ma = make_abi('float64', steps_per_launch=60)
  eng = Engine(ca, ma, n)
  rng = np.random.default_rng(99)
  params = np.zeros((n, 4)); params[:, 0] = ca.lateral_friction; params[:, 1] = 1.0
  if name == 'randomised':
    params[:, 0] = rng.uniform(0.3, 1.0, n)
    params[:, 1] = rng.uniform(0.8, 1.2, n)
    eng.set_params(abi.PARAM_FRICTION, torch.as_tensor(params[:, 0], device='cuda'))
    eng.set_params(abi.PARAM_BASE_MASS_SCALE, torch.as_tensor(params[:, 1], device='cuda'))
    eng.settle()
  ph = so.OraclePhysics(ca, ma)
  st = eng.state.cpu().numpy().copy()
  a = np.stack([random_actions(rng, n) for _ in range(60)])
  eng.rollout(torch.as_tensor(a, device='cuda'), abi.STEP_PHYSICS)
  for k in range(60):
    ph.step(st, a[k], params, threads=threads)
  got = eng.state.cpu().numpy()
  err = np.abs(got[:, :29] - st[:, :29]).max(axis=1)
  print('%s n=%d, 60 steps of U(-2pi, 2pi) targets: engine vs oracle max %.1e, p99 %.1e, median %.1e' % (name, n, err.max(), np.quantile(err, 0.99), np.median(err)))
  assert np.isfinite(got[:, :29]).all() and err.max() < 1e-9, err.max()
  eng.close()
