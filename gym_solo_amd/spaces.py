"""gym / gymnasium soft import with a minimal stand-in.

The reference depends on ``gym`` (setup.py:7) for ``gym.Env`` and ``gym.spaces.Box`` only
(solo8_base_env.py:17, obs.py:11, solo8v2vanilla.py:9-10).  Neither gym nor gymnasium is
installed in the build image, so a tiny ``Box``/``Env`` with the attributes the reference's
code and tests touch (low/high as float32, shape, sample, contains, is_bounded, ==) is used
when they are absent.
"""
import numpy as np

try:  # pragma: no cover - not available in the build image
  import gym as _gym
  from gym import spaces as _spaces
  Env = _gym.Env
  Box = _spaces.Box
  Space = _spaces.Space
  HAVE_GYM = True
except Exception:  # noqa: BLE001
  try:  # pragma: no cover
    import gymnasium as _gym
    from gymnasium import spaces as _spaces
    Env = _gym.Env
    Box = _spaces.Box
    Space = _spaces.Space
    HAVE_GYM = True
  except Exception:  # noqa: BLE001
    HAVE_GYM = False

    class Space:
      pass

    class Env:
      metadata = {}

    class Box(Space):
      def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is not None:
          low = np.full(shape, low, dtype=dtype)
          high = np.full(shape, high, dtype=dtype)
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        if self.low.shape != self.high.shape:
          raise ValueError('low and high must have the same shape')
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)

      def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)

      def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

      def is_bounded(self):
        return bool(np.all(np.isfinite(self.low)) and np.all(np.isfinite(self.high)))

      def __eq__(self, other):
        return (isinstance(other, Box) and self.shape == other.shape
                and np.allclose(self.low, other.low) and np.allclose(self.high, other.high))

      def __repr__(self):
        return 'Box({}, {}, {}, {})'.format(self.low.min(), self.high.max(), self.shape, self.dtype)
