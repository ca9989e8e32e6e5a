"""ctypes driver of tests/emu/libsolo_emu.so: the PRODUCT kernel source compiled for the CPU
fibre emulator (test infrastructure; see tests/emu/wave_emu.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

from gym_solo_amd import abi

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu')


def load(variant=''):
  name = 'libsolo_emu%s.so' % (('_' + variant) if variant else '')
  path = os.path.join(_DIR, name)
  deps = [os.path.join(_DIR, f) for f in ('emu_harness.cpp', 'wave_emu.h')] + [
    os.path.join(_DIR, '..', '..', 'gym_solo_amd', 'csrc', f)
    for f in ('solo_step_kernel.h', 'solo_kernel_params.h', 'solo_outputs.h')]
  if not os.path.exists(path) or any(os.path.getmtime(d) > os.path.getmtime(path) for d in deps):
    subprocess.check_call(['make', '-s', '-C', _DIR, variant or 'all'])
  lib = C.CDLL(path)
  dp = C.POINTER(C.c_double)
  lib.solo_emu_step.restype = C.c_int
  lib.solo_emu_step.argtypes = [C.POINTER(abi.SoloConfig), C.POINTER(abi.SoloModel), C.c_void_p,
                                C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, C.c_void_p,
                                C.c_void_p, dp, C.c_uint32, C.c_void_p, dp]
  lib.solo_emu_rollout.restype = C.c_int
  lib.solo_emu_rollout.argtypes = [C.POINTER(abi.SoloConfig), C.POINTER(abi.SoloModel), C.c_void_p,
                                   C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, C.c_void_p,
                                   C.c_void_p, dp, C.c_uint32, C.c_void_p, dp]
  lib.solo_emu_last_cost.restype = C.c_int
  lib.solo_emu_last_cost.argtypes = [C.c_void_p, C.c_int]
  lib.solo_emu_take_fault.restype = C.c_int      # the fault word a wave sets when it gives up waiting (KBuffers::fault); reading clears it
  lib.solo_emu_sabotage_queue.argtypes = [C.c_int]
  return lib


def _dp(a):
  assert a.dtype == np.float64 and a.flags['C_CONTIGUOUS']
  return a.ctypes.data_as(C.POINTER(C.c_double))


class EmuEngine:
  """Same call shape as gym_solo_amd.engine.Engine, numpy buffers, CPU emulation."""

  def __init__(self, cfg, model, n, program=None, variant='', terrain=None):
    self.lib = load(variant)
    self.terrain = terrain
    self.cfg, self.model, self.n = cfg, model, n
    self.program = program
    self.state = np.zeros((n, abi.STATE_STRIDE))
    self.state[:, abi.S_POS:abi.S_POS + 3] = list(cfg.start_pos)
    self.state[:, abi.S_QUAT:abi.S_QUAT + 4] = list(cfg.start_quat)
    self.snapshot = self.state.copy()
    self.targets = np.zeros((n, abi.NUM_JOINTS))
    self.params = np.zeros((n, 4))
    self.params[:, 0] = cfg.lateral_friction
    self.params[:, 1] = 1.0
    d = program.num_obs if program is not None else 0
    self.obs = np.zeros((n, max(d, 1)))
    self.reward = np.zeros(n)
    self.done = np.zeros(n, dtype=np.uint8)
    self.term_count = np.zeros((n, abi.MAX_TERMS), dtype=np.int32)
    self.stats = np.zeros((abi.STATS_SHARDS, abi.STATS_WIDTH))
    self.warm = np.zeros((n, 64))   # the warm-start cache (SoloConfig.solver_warm_start)

  def step(self, actions=None, flags=abi.STEP_ALL):
    if self.program is None:
      flags &= abi.STEP_PHYSICS
    if flags == 0:
      return
    a = None
    if actions is not None:
      a = np.ascontiguousarray(actions, dtype=np.float64)
    rc = self.lib.solo_emu_step(
      C.byref(self.cfg), C.byref(self.model),
      C.cast(C.pointer(self.program), C.c_void_p) if self.program is not None else None,
      self.cfg.dtype, self.n, _dp(self.state), _dp(self.snapshot),
      _dp(a) if a is not None else None, _dp(self.targets), _dp(self.params), _dp(self.obs),
      _dp(self.reward), self.done.ctypes.data, self.term_count.ctypes.data, _dp(self.stats), flags,
      C.byref(self.terrain) if getattr(self, 'terrain', None) is not None else None, _dp(self.warm))
    if rc:
      raise RuntimeError('emu step failed: %d' % rc)

  @property
  def cost(self):
    """Gauss-Seidel sweeps each robot ran in the last launch (the engine's view.cost)."""
    out = np.zeros(self.n, dtype=np.int32)
    self.lib.solo_emu_last_cost(out.ctypes.data, self.n)
    return out

  def rollout(self, actions, flags=abi.STEP_ALL):
    """One fused multi-step launch: actions [K, N, 12] -> (obs [K,N,D], reward [K,N], done [K,N])."""
    a = np.ascontiguousarray(actions, dtype=np.float64)
    k = a.shape[0]
    d = self.program.num_obs if self.program is not None else 0
    obs = np.zeros((k, self.n, max(d, 1)))
    rew = np.zeros((k, self.n))
    done = np.zeros((k, self.n), dtype=np.uint8)
    rc = self.lib.solo_emu_rollout(
      C.byref(self.cfg), C.byref(self.model),
      C.cast(C.pointer(self.program), C.c_void_p) if self.program is not None else None,
      self.cfg.dtype, self.n, k, _dp(self.state), _dp(self.snapshot), _dp(a), _dp(self.targets),
      _dp(self.params), _dp(obs), _dp(rew), done.ctypes.data, self.term_count.ctypes.data,
      _dp(self.stats), flags, C.byref(self.terrain) if getattr(self, 'terrain', None) is not None else None, _dp(self.warm))
    if rc:
      raise RuntimeError('emu rollout failed: %d' % rc)
    return obs, rew, done

  def settle(self):
    tg = np.tile(np.array(list(self.cfg.settle_targets)), (self.n, 1)) / self.cfg.action_scale
    self.state[:] = 0
    self.state[:, abi.S_POS:abi.S_POS + 3] = list(self.cfg.start_pos)
    self.state[:, abi.S_QUAT:abi.S_QUAT + 4] = list(self.cfg.start_quat)
    self.warm[:] = 0
    for _ in range(self.cfg.settle_steps):
      self.step(tg, abi.STEP_PHYSICS)
    self.snapshot[:] = self.state
    self.warm[:] = 0   # (the snapshot starts from an empty warm-start cache, as Engine<T>::settle leaves it)


class EmuTorchEngine:
  """Drop-in for gym_solo_amd.engine.Engine backed by the CPU emulator: torch CPU tensors that
  share memory with the emulator's numpy buffers.  Lets the host API (envs, factories, client
  facade) be tested without a GPU while still running the product kernel source."""

  _settled = {}  # (config bytes, n) -> snapshot: the settle loop is deterministic

  def __init__(self, cfg, model, n, device=0):
    import torch
    import ctypes as C
    self._torch = torch
    self._e = EmuEngine(cfg, model, n)
    key = (bytes(C.string_at(C.addressof(cfg), C.sizeof(cfg))), n)
    if key not in EmuTorchEngine._settled:
      self._e.settle()
      EmuTorchEngine._settled[key] = self._e.snapshot.copy()
    self._e.snapshot[:] = EmuTorchEngine._settled[key]
    self._e.state[:] = self._e.snapshot
    self._e.targets[:] = self._settle_targets(cfg)  # as Engine<T>::settle leaves them
    self.cfg, self.model = cfg, model
    self.num_envs, self.device = n, device
    self.tdtype = torch.float64  # emulator buffers are double; kernel arithmetic is cfg.dtype
    self.program = None
    self.obs_dim = 0
    for name in ('state', 'snapshot', 'targets', 'params', 'reward', 'done', 'term_count'):
      setattr(self, name, torch.from_numpy(getattr(self._e, name)))
    self.stats_shards = torch.from_numpy(self._e.stats)
    self.done_bool = self.done.view(torch.bool)
    self.obs = None

  def set_program(self, program):
    if (program.num_obs > abi.MAX_OBS or program.num_reward_ops > abi.MAX_REWARD_OPS
        or program.num_terms > abi.MAX_TERMS):
      raise ValueError('program too large')
    self.program = program
    self._e.program = program
    self._e.obs = np.zeros((self.num_envs, max(program.num_obs, 1)))
    self.obs_dim = program.num_obs
    self.obs = self._torch.from_numpy(self._e.obs) if program.num_obs else None

  def step(self, actions=None, flags=abi.STEP_ALL):
    prog = self.program
    if flags & (abi.STEP_OBS | abi.STEP_REWARD | abi.STEP_DONE):
      # same checks as Engine<T>::check_flags in solo_engine.hip
      if prog is None:
        raise ValueError('no observation/reward/termination program registered')
      if (flags & abi.STEP_OBS) and prog.num_obs == 0:
        raise ValueError('Need to register at least one observation instance')
      if (flags & abi.STEP_REWARD) and prog.num_reward_ops == 0:
        raise ValueError('Need to register at least one reward instance')
      if (flags & abi.STEP_DONE) and prog.num_terms == 0:
        raise ValueError('Need to register at least one termination instance')
    a = None if actions is None else actions.detach().cpu().numpy()
    self._e.step(a, flags)

  def set_targets(self, actions):
    self._e.targets[:] = actions.detach().cpu().numpy() * self.cfg.action_scale

  # ---- the rollout calls of gym_solo_amd.engine.Engine (what bench.py drives) ------------------------------------
  @property
  def steps_per_launch(self):
    return max(1, int(self.cfg.steps_per_launch))

  def plan(self, num_steps):
    """The engine's launch policy as far as an emulator has one (solo_engine.hip: make_plan): fused launches of
    min(K, 250) steps when the choice is left to the engine; one chain, no migration unless configured."""
    spl = min(int(num_steps), 250) if int(self.cfg.steps_per_launch) == -1 else min(self.steps_per_launch, int(num_steps))
    spl = max(1, spl)
    return {'steps_per_launch': spl, 'launches': -(-int(num_steps) // spl), 'slices': 1, 'migrate_steps': max(0, int(self.cfg.migrate_steps)),
            'waves_per_simd': 0, 'resident_robots': 0}

  def time_rollout(self, actions, flags=abi.STEP_ALL, out=None):
    import time
    t0 = time.perf_counter()
    self.rollout(actions, flags, out=out)
    return (time.perf_counter() - t0) * 1e3 / self.plan(actions.shape[0])['launches']

  def rollout_buffers(self, k):
    torch = self._torch
    return (torch.empty(k, self.num_envs, max(self.obs_dim, 1), dtype=torch.float64), torch.empty(k, self.num_envs, dtype=torch.float64),
            torch.empty(k, self.num_envs, dtype=torch.uint8))

  def rollout(self, actions, flags=abi.STEP_ALL, record=False, out=None):
    """K open-loop steps in fused launches of steps_per_launch steps (robot migration as configured)."""
    a = actions.detach().cpu().numpy()
    k = a.shape[0]
    spl = self.plan(k)['steps_per_launch']
    parts = [self._e.rollout(a[i:i + spl], flags) for i in range(0, k, spl)]
    if not record and out is None:
      return None
    obs, rew, done = out if out is not None else self.rollout_buffers(k)
    torch = self._torch
    obs.copy_(torch.from_numpy(np.concatenate([p[0] for p in parts])))
    rew.copy_(torch.from_numpy(np.concatenate([p[1] for p in parts])))
    done.copy_(torch.from_numpy(np.concatenate([p[2] for p in parts])))
    return obs, rew, done

  def time_step(self, actions=None, flags=abi.STEP_ALL, reps=100):
    """Mean ms per emulated launch (wall clock: there is no device to time)."""
    import time
    spl = self.steps_per_launch
    reps = int(actions.shape[0]) // spl
    t0 = time.perf_counter()
    self.rollout(actions[:reps * spl], flags)
    return (time.perf_counter() - t0) * 1e3 / max(reps, 1)

  @property
  def cost(self):
    return self._torch.from_numpy(self._e.cost)

  @staticmethod
  def _settle_targets(cfg):
    return np.array(list(cfg.settle_targets))

  def reset(self, mask=None):
    """solo_reset_kernel: snapshot, termination counters and the settle pose as motor targets."""
    e = self._e
    m = slice(None) if mask is None else mask.detach().cpu().numpy().astype(bool)
    e.state[m] = e.snapshot[m]
    e.term_count[m] = 0
    e.warm[m] = 0
    e.targets[m] = self._settle_targets(self.cfg)

  def settle(self):
    self._e.settle()

  def set_params(self, which, per_env):
    self._e.params[:, which] = per_env.detach().cpu().numpy()

  def synchronize(self):
    pass

  def close(self):
    pass

  @property
  def stats(self):
    return self.stats_shards.sum(dim=0)

  @property
  def kernel_name(self):
    return 'emulated'


def make_emu_env_class():
  """Solo8VanillaEnv on the emulator engine (the host-side API without a GPU): tests/test_env_host.py, and bench.py's
  CPU rehearsal (SOLO_BENCH_ENGINE=emu)."""
  from gym_solo_amd.core.configs import config_to_abi
  from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaEnv
  from gym_solo_amd.model import JOINT_NAMES

  class EmuSolo8VanillaEnv(Solo8VanillaEnv):
    def create_engine(self):
      cfg = config_to_abi(self.config, self.config.starting_joint_pos, JOINT_NAMES, normalize_actions=self._normalize)
      return EmuTorchEngine(cfg, self.solo_model.to_abi(), self.config.num_envs)

  return EmuSolo8VanillaEnv
