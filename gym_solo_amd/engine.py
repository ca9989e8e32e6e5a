"""ctypes binding of the HIP engine (``gym_solo_amd/csrc/libsolo_hip.so``, C-ABI in
``include/solo_engine.h``).

PyTorch is plumbing here: it provides the device tensors handed IN (actions, masks, per-env
parameters) and zero-copy views of the engine-OWNED buffers (``__cuda_array_interface__``).
There is NO CPU fallback: if the library is missing or no MI355X is visible, this raises.
"""
import ctypes as C
import functools
import os

import numpy as np

from gym_solo_amd import abi

_LIB_PATH = os.environ.get('SOLO_HIP_LIB') or os.path.join(
  os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libsolo_hip.so')


class EngineError(RuntimeError):
  pass


@functools.lru_cache(maxsize=None)
def load_library(path=_LIB_PATH):
  if not os.path.exists(path):
    raise EngineError(
      'HIP engine library not found at {} — build it with `make -C gym_solo_amd/csrc` or '
      '`python -c "import __graft_entry__ as g; g.build()"`. There is no CPU fallback.'.format(path))
  # torch first: it carries its own libamdhip64 / libhsa-runtime64 (same sonames as /opt/rocm's), and
  # the process must end up with ONE HIP runtime - the one whose streams and tensors the engine is
  # handed.  Loaded the other way round, the second runtime sees no device.
  import torch  # noqa: F401
  lib = C.CDLL(path)
  abi.bind(lib)
  if lib.solo_abi_version() != abi.ABI_VERSION:
    raise EngineError('ABI version mismatch between libsolo_hip.so and gym_solo_amd.abi')
  return lib


class _DeviceArray:
  """Borrowed view of engine-owned device memory for ``torch.as_tensor`` (zero copy).  It holds no
  reference to the engine (torch keeps this object alive from C++, which Python's GC cannot see
  through: a back-reference would make every engine immortal)."""

  def __init__(self, ptr, shape, typestr):
    self.__cuda_array_interface__ = {
      'shape': tuple(int(s) for s in shape), 'typestr': typestr, 'data': (int(ptr), False),
      'version': 2, 'strides': None}


def _raise(rc, msg):
  if rc in (abi.ERR_INVALID_ARG, abi.ERR_NO_PROGRAM, abi.ERR_UNSUPPORTED_MODEL):
    # misuse maps to ValueError exactly where the reference raises it
    raise ValueError(msg)
  raise EngineError('{} (status {})'.format(msg, rc))


def _destroy(lib, handle, torch, device):
  try:
    torch.cuda.synchronize(device)
  except Exception:  # noqa: BLE001 - interpreter shutdown
    pass
  lib.solo_engine_destroy(handle)


class Engine:
  """One handle per GPU/process; not thread-safe; stream-ordered on torch's current stream.

  Lifetime: the device buffers belong to the engine and are freed by ``close()`` (or when the
  Engine object is collected).  The tensors it hands out (``state``, ``obs``, ``reward`` ... and
  the zero-copy outputs of ``Solo8VanillaEnv(copy_outputs=False)``) ALIAS those buffers: keep the
  engine alive while they are in use, and do not touch them after ``close()``."""

  def __init__(self, cfg: abi.SoloConfig, model: abi.SoloModel, num_envs: int, device: int = 0):
    import torch  # plumbing only
    self._torch = torch
    self.lib = load_library()
    self.cfg, self.model = cfg, model
    self.num_envs, self.device = int(num_envs), int(device)
    self.tdtype = torch.float32 if cfg.dtype == abi.F32 else torch.float64
    self._h = C.c_void_p()
    rc = self.lib.solo_engine_create(C.byref(cfg), C.byref(model), self.num_envs, self.device,
                                     C.byref(self._h))
    if rc != abi.OK:
      self._h = None
      _raise(rc, 'solo_engine_create: ' + self.lib.solo_last_create_error().decode())
    import weakref
    self._finalizer = weakref.finalize(self, _destroy, self.lib, self._h, torch, self.device)
    self.program = None
    self._action_shape = (self.num_envs, abi.NUM_JOINTS)
    self._make_views()

  # ---- plumbing --------------------------------------------------------------------------
  @property
  def is_closed(self):
    return self._h is None

  def _handle(self):
    if self._h is None:
      raise EngineError('the engine was closed (its device buffers are freed)')
    return self._h

  def _check(self, rc, what):
    if rc != abi.OK:
      _raise(rc, '{}: {}'.format(what, self.lib.solo_engine_last_error(self._handle()).decode()))

  def _stream(self):
    return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

  def _make_views(self):
    torch = self._torch
    v = abi.SoloStateView()
    self._check(self.lib.solo_engine_get_view(self._handle(), C.byref(v)), 'get_view')
    real = '<f4' if v.dtype == abi.F32 else '<f8'
    n = v.num_envs
    dev = 'cuda:%d' % self.device
    def view(ptr, shape, typestr):
      return torch.as_tensor(_DeviceArray(ptr, shape, typestr), device=dev)
    self.state = view(v.state, (n, abi.STATE_STRIDE), real)
    self.snapshot = view(v.snapshot, (n, abi.STATE_STRIDE), real)
    self.targets = view(v.targets, (n, abi.NUM_JOINTS), real)
    self.reward = view(v.reward, (n,), real)
    self.done = view(v.done, (n,), '|u1')
    self.done_bool = self.done.view(torch.bool)  # zero-copy bool alias of the uint8 flags
    self.term_count = view(v.term_count, (n, abi.MAX_TERMS), '<i4')
    self.params = view(v.params, (n, 4), real)
    self.stats_shards = view(v.stats, (abi.STATS_SHARDS, abi.STATS_WIDTH), '<f8')
    self.cost = view(v.cost, (n,), '<i4')
    self.warm = view(v.warm, (n, 64), real)   # the warm-start cache (SoloConfig.solver_warm_start); zeros while off
    self.obs_dim = v.obs_dim
    self._obs_ptr, self._real = v.obs, real
    self.obs = view(v.obs, (n, max(v.obs_dim, 1)), real) if v.obs_dim else None

  def _dev_ptr(self, t, shape, dtype, name):
    torch = self._torch
    if not isinstance(t, torch.Tensor):
      raise ValueError('{} must be a torch tensor on cuda:{}'.format(name, self.device))
    if not t.is_cuda or t.device.index != self.device:
      raise ValueError('{} must live on cuda:{}'.format(name, self.device))
    if t.dtype != dtype:
      raise ValueError('{} must have dtype {}'.format(name, dtype))
    if tuple(t.shape) != tuple(shape):
      raise ValueError('{} must have shape {}, got {}'.format(name, tuple(shape), tuple(t.shape)))
    if not t.is_contiguous():
      raise ValueError('{} must be contiguous'.format(name))
    return C.c_void_p(t.data_ptr())

  def _as_real(self, t):
    """Actions / per-robot parameters in another floating precision than the engine's are converted here (one cast
    kernel on the caller's stream); the C-ABI itself takes the engine's precision only."""
    torch = self._torch
    if isinstance(t, torch.Tensor) and t.is_floating_point() and t.dtype != self.tdtype:
      return t.to(self.tdtype)
    return t

  # ---- C-ABI calls -----------------------------------------------------------------------
  def set_program(self, program: abi.SoloProgram):
    self._check(self.lib.solo_engine_set_program(self._handle(), C.byref(program)), 'set_program')
    self.program = program
    self._make_views()

  def reset(self, mask=None):
    p = None
    if mask is not None:
      p = self._dev_ptr(mask, (self.num_envs,), self._torch.uint8, 'mask')
    self._check(self.lib.solo_engine_reset(self._handle(), p, self._stream()), 'reset')

  def settle(self):
    self._check(self.lib.solo_engine_settle(self._handle(), self._stream()), 'settle')

  def set_targets(self, actions):
    actions = self._as_real(actions)
    p = self._dev_ptr(actions, (self.num_envs, abi.NUM_JOINTS), self.tdtype, 'actions')
    self._check(self.lib.solo_engine_set_targets(self._handle(), p, self._stream()), 'set_targets')

  def step(self, actions=None, flags=abi.STEP_ALL):
    p = None
    if actions is not None:
      actions = self._as_real(actions)
      p = self._dev_ptr(actions, self._action_shape, self.tdtype, 'actions')
    rc = self.lib.solo_engine_step(self._handle(), p, flags, self._stream())
    if rc != abi.OK:
      self._check(rc, 'step')

  @property
  def steps_per_launch(self):
    """The CONFIGURED steps per launch (1 when the choice is left to the engine: plan(k) says what a rollout of k steps runs with)."""
    return max(1, int(self.cfg.steps_per_launch))

  def plan(self, num_steps):
    """The launch geometry a rollout of num_steps steps runs with (solo_engine_plan): dict(steps_per_launch, launches,
    slices, migrate_steps, waves_per_simd, resident_robots) - the measured launch policy lives in the engine."""
    p = abi.SoloLaunchPlan()
    self._check(self.lib.solo_engine_plan(self._handle(), int(num_steps), C.byref(p)), 'plan')
    return {name: int(getattr(p, name)) for name, _ in abi.SoloLaunchPlan._fields_}

  def reserve(self, num_steps, flags=abi.STEP_ALL):
    """Sizes the lazily grown scratch (the record scratch of fused launches, the migration queues) NOW for rollouts of up
    to num_steps steps (solo_engine_reserve): the first rollout of a larger geometry otherwise synchronises the device and
    re-allocates - not legal inside a HIP graph capture, and where an out-of-memory would surface mid-run."""
    self._check(self.lib.solo_engine_reserve(self._handle(), int(num_steps), flags), 'reserve')

  def time_rollout(self, actions, flags=abi.STEP_ALL, out=None):
    """Mean ms per LAUNCH of one rollout of actions.shape[0] steps run exactly as rollout() runs it (plan()'s
    geometry; out = rollout_buffers(K): every step's outputs recorded, as a rollout collector's call does), measured with
    HIP events on the launch streams."""
    actions = self._as_real(actions)
    k = int(actions.shape[0])
    p = self._dev_ptr(actions, (k, self.num_envs, abi.NUM_JOINTS), self.tdtype, 'actions')
    outs = [None, None, None]
    if out is not None:
      obs, rew, done = out
      self._dev_ptr(obs, (k, self.num_envs, max(self.obs_dim, 1)), self.tdtype, 'obs_out')
      self._dev_ptr(rew, (k, self.num_envs), self.tdtype, 'reward_out')
      self._dev_ptr(done, (k, self.num_envs), self._torch.uint8, 'done_out')
      outs = [C.c_void_p(t.data_ptr()) for t in out]
    ms = C.c_double()
    self._check(self.lib.solo_engine_time_rollout(self._handle(), p, k, flags, outs[0], outs[1], outs[2], self._stream(), C.byref(ms)), 'time_rollout')
    return ms.value

  def rollout_buffers(self, k):
    """Trajectory buffers for rollout(record=True): obs [K,N,D], reward [K,N], done [K,N] uint8."""
    torch = self._torch
    dev = self.state.device
    return (torch.empty(k, self.num_envs, max(self.obs_dim, 1), device=dev, dtype=self.tdtype),
            torch.empty(k, self.num_envs, device=dev, dtype=self.tdtype),
            torch.empty(k, self.num_envs, device=dev, dtype=torch.uint8))

  def rollout(self, actions, flags=abi.STEP_ALL, record=False, out=None):
    """K open-loop env steps with actions [K, N, 12]; ceil(K / steps_per_launch) fused launches.
    record=True (or out=rollout_buffers(K)) keeps every step's (obs [K,N,D], reward [K,N],
    done [K,N] uint8) — what a rollout collector reads; the engine's view holds the last step's
    outputs afterwards in either case."""
    torch = self._torch
    actions = self._as_real(actions)
    k = int(actions.shape[0])
    p = self._dev_ptr(actions, (k, self.num_envs, abi.NUM_JOINTS), self.tdtype, 'actions')
    if not record and out is None:
      self._check(self.lib.solo_engine_rollout(self._handle(), p, k, flags, self._stream()), 'rollout')
      return None
    obs, rew, done = out if out is not None else self.rollout_buffers(k)
    self._dev_ptr(obs, (k, self.num_envs, max(self.obs_dim, 1)), self.tdtype, 'obs_out')
    self._dev_ptr(rew, (k, self.num_envs), self.tdtype, 'reward_out')
    self._dev_ptr(done, (k, self.num_envs), torch.uint8, 'done_out')
    self._check(self.lib.solo_engine_rollout_record(
      self._handle(), p, k, flags, C.c_void_p(obs.data_ptr()), C.c_void_p(rew.data_ptr()),
      C.c_void_p(done.data_ptr()), self._stream()), 'rollout_record')
    return obs, rew, done

  def time_step(self, actions=None, flags=abi.STEP_ALL, reps=100):
    """Mean ms per LAUNCH of the step kernel (each launch fuses steps_per_launch env steps),
    measured with HIP events on the launch stream.  actions: [reps * steps_per_launch, N, 12]
    (fresh actions every step) or None."""
    p = None
    if actions is not None:
      spl = self.steps_per_launch
      reps = int(actions.shape[0]) // spl
      p = self._dev_ptr(actions[:reps * spl], (reps * spl, self.num_envs, abi.NUM_JOINTS),
                        self.tdtype, 'actions')
    ms = C.c_double()
    self._check(self.lib.solo_engine_time_step(self._handle(), p, flags, reps, self._stream(),
                                               C.byref(ms)), 'time_step')
    return ms.value

  def set_terrain(self, terrain):
    """terrain: abi.SoloTerrain (abi.make_terrain(heights, cell)) or None for the flat plane;
    re-settles (the reset snapshot depends on the ground)."""
    self._check(self.lib.solo_engine_set_terrain(
      self._handle(), C.byref(terrain) if terrain is not None else None, self._stream()), 'set_terrain')

  def set_order(self, order=None):
    """Launch order of the robots: int32 [N] permutation (workgroup b steps robot order[b]) or None =
    identity.  With rollout slices every slice's positions must hold that slice's own robots."""
    p = None
    if order is not None:
      p = self._dev_ptr(order, (self.num_envs,), self._torch.int32, 'order')
    self._check(self.lib.solo_engine_set_order(self._handle(), p, self._stream()), 'set_order')

  def balance(self):
    """Cost-balanced scheduling: dispatch the costliest robots first (Gauss-Seidel sweeps of the last
    launch, persistent within an episode), slice by slice.  Matters when N exceeds the chip's 4096
    resident waves; results do not depend on it."""
    torch = self._torch
    g = int(self.cfg.rollout_streams)
    g = 2 if g == abi.AUTO else max(1, g)   # (left to the engine, a rollout runs on one or two slices: an order that respects two respects one)
    g = g if (g > 1 and self.num_envs >= 2 * g) else 1
    parts = []
    for s in range(g):
      lo, hi = self.num_envs * s // g, self.num_envs * (s + 1) // g
      parts.append(torch.argsort(self.cost[lo:hi], descending=True, stable=True).to(torch.int32) + lo)
    self.set_order(torch.cat(parts).contiguous())

  def set_params(self, which, per_env):
    per_env = self._as_real(per_env)
    p = self._dev_ptr(per_env, (self.num_envs,), self.tdtype, 'per_env')
    self._check(self.lib.solo_engine_set_params(self._handle(), which, p, self._stream()), 'set_params')

  @property
  def stats(self):
    """[sum return, sum return^2, episodes, sum length, -, diverged, -, -] (float64, summed over
    the shards the kernel accumulates into)."""
    return self.stats_shards.sum(dim=0)

  @property
  def kernel_name(self):
    return self.lib.solo_engine_kernel_name(self._handle()).decode()

  def synchronize(self):
    self._torch.cuda.synchronize(self.device)

  # ---- checkpoint / resume (the reference has none - reset() rebuilds the world; here the
  #      whole simulation is a handful of device tensors) --------------------------------------------------------
  _CHECKPOINT = ('state', 'snapshot', 'targets', 'term_count', 'params', 'stats_shards', 'cost', 'warm')
  CHECKPOINT_VERSION = 2   # 1: before ABI 4 (no snapshot, no warm-start cache, no version tag); 2: the fields above + 'version'

  def get_state(self):
    """Everything a run continues from, as clones on the device: the robots' state records (episodic return / length
    accumulators included), the reset snapshot every auto-reset restores (it reflects whatever parameters and terrain
    were in force at the last settle() / set_terrain(), NOT the current ones: set_params does not re-settle), the
    commanded motor targets, the TimeBased counters, the per-robot parameters, the episodic statistics and the
    per-robot solver cost (which decides a closed-loop launch's wave priorities, not results).  Terrain and the
    compiled configuration are NOT part of it: restore into an engine built with the same ones."""
    self._torch.cuda.current_stream(self.device).synchronize()
    ck = {name: getattr(self, name).clone() for name in self._CHECKPOINT}
    ck['version'] = self.CHECKPOINT_VERSION
    return ck

  def set_state(self, checkpoint):
    """Restores a get_state() checkpoint (of an engine with the same number of robots and precision): the next step
    continues bit for bit where the checkpointed run would have.  Listeners registered with on_restore() (the env's
    client: its cached observations / rewards are those of the state before the restore) are told."""
    version = int(checkpoint.get('version', 1))
    if version > self.CHECKPOINT_VERSION:
      raise ValueError('checkpoint version {} is newer than this engine understands ({})'.format(version, self.CHECKPOINT_VERSION))
    # version 1 (before ABI 4) carries neither the reset snapshot nor the warm-start cache: the engine's CURRENT snapshot
    # stays in force (it is what reset() would restore anyway after the same settle()) and the cache starts empty
    optional = ('snapshot', 'warm') if version < 2 else ()
    missing = [name for name in self._CHECKPOINT if name not in checkpoint and name not in optional]
    if missing:
      raise ValueError('checkpoint (version {}) lacks the field(s) {}'.format(version, missing))
    names = [name for name in self._CHECKPOINT if name in checkpoint]
    for name in names:
      src, dst = checkpoint[name], getattr(self, name)
      if tuple(src.shape) != tuple(dst.shape) or src.dtype != dst.dtype:
        raise ValueError('checkpoint field {!r} has shape {} / dtype {}, the engine has {} / {}'.format(
          name, tuple(src.shape), src.dtype, tuple(dst.shape), dst.dtype))
    for name in names:
      getattr(self, name).copy_(checkpoint[name])
    if 'warm' not in checkpoint:
      self.warm.zero_()
    for hook in getattr(self, '_restore_hooks', ()):
      hook()

  def on_restore(self, hook):
    """hook() is called after every set_state()."""
    if not hasattr(self, '_restore_hooks'):
      self._restore_hooks = []
    self._restore_hooks.append(hook)

  def close(self):
    """solo_engine_destroy: frees every device buffer.  Tensors handed out earlier dangle."""
    if getattr(self, '_h', None):
      for name in ('state', 'snapshot', 'targets', 'reward', 'done', 'done_bool', 'term_count', 'params',
                   'stats_shards', 'obs', 'cost', 'warm'):
        setattr(self, name, None)
      self._finalizer()  # synchronises the device, then destroys the handle (runs at most once)
      self._h = None
