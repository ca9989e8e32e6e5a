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
    for f in ('solo_step_kernel.h', 'solo_kernel_params.h')]
  if not os.path.exists(path) or any(os.path.getmtime(d) > os.path.getmtime(path) for d in deps):
    subprocess.check_call(['make', '-s', '-C', _DIR, variant or 'all'])
  lib = C.CDLL(path)
  dp = C.POINTER(C.c_double)
  lib.solo_emu_step.restype = C.c_int
  lib.solo_emu_step.argtypes = [C.POINTER(abi.SoloConfig), C.POINTER(abi.SoloModel), C.c_void_p,
                                C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, C.c_void_p,
                                C.c_void_p, dp, C.c_uint32]
  return lib


def _dp(a):
  assert a.dtype == np.float64 and a.flags['C_CONTIGUOUS']
  return a.ctypes.data_as(C.POINTER(C.c_double))


class EmuEngine:
  """Same call shape as gym_solo_amd.engine.Engine, numpy buffers, CPU emulation."""

  def __init__(self, cfg, model, n, program=None, variant=''):
    self.lib = load(variant)
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
    self.stats = np.zeros(8)

  def step(self, actions=None, flags=abi.STEP_ALL):
    if self.program is None:
      flags &= abi.STEP_PHYSICS
    a = None
    if actions is not None:
      a = np.ascontiguousarray(actions, dtype=np.float64)
    rc = self.lib.solo_emu_step(
      C.byref(self.cfg), C.byref(self.model),
      C.cast(C.pointer(self.program), C.c_void_p) if self.program is not None else None,
      self.cfg.dtype, self.n, _dp(self.state), _dp(self.snapshot),
      _dp(a) if a is not None else None, _dp(self.targets), _dp(self.params), _dp(self.obs),
      _dp(self.reward), self.done.ctypes.data, self.term_count.ctypes.data, _dp(self.stats), flags)
    if rc:
      raise RuntimeError('emu step failed: %d' % rc)

  def settle(self):
    tg = np.tile(np.array(list(self.cfg.settle_targets)), (self.n, 1)) / self.cfg.action_scale
    self.state[:] = 0
    self.state[:, abi.S_POS:abi.S_POS + 3] = list(self.cfg.start_pos)
    self.state[:, abi.S_QUAT:abi.S_QUAT + 4] = list(self.cfg.start_quat)
    for _ in range(self.cfg.settle_steps):
      self.step(tg, abi.STEP_PHYSICS)
    self.snapshot[:] = self.state
