"""gym_solo_amd — MI355X-native batched replacement for the hot path of WPI-MMR/gym_solo.

See DESIGN.md. Import is light on purpose: the HIP engine is loaded lazily by
``gym_solo_amd.engine`` and fails loudly if ``libsolo_hip.so`` is missing.

Environment ids (gym_solo/__init__.py:3-11): ``solo8vanilla-v0`` is registered with gym /
gymnasium when one of them is installed, and is always available through ``gym_solo_amd.make``.
``solo8vanilla-realtime-v0`` (wall-clock GUI env) is out of scope for the batched engine.
"""
__version__ = '0.1.0'

_REGISTRY = {'solo8vanilla-v0': 'gym_solo_amd.envs:Solo8VanillaEnv'}


def make(env_id, **kwargs):
  """``gym.make`` stand-in: ``gym_solo_amd.make('solo8vanilla-v0', config=cfg)``."""
  if env_id == 'solo8vanilla-realtime-v0':
    raise ValueError('the realtime (wall-clock, GUI) env is out of scope for the batched engine')
  if env_id not in _REGISTRY:
    raise ValueError('unknown environment id {!r} (known: {})'.format(env_id, sorted(_REGISTRY)))
  import importlib
  module, cls = _REGISTRY[env_id].split(':')
  return getattr(importlib.import_module(module), cls)(**kwargs)


def _register():
  for mod in ('gym', 'gymnasium'):
    try:
      registration = __import__(mod + '.envs.registration', fromlist=['register'])
    except Exception:  # noqa: BLE001 - neither package is installed in the build image
      continue
    for env_id, entry in _REGISTRY.items():
      try:
        registration.register(id=env_id, entry_point=entry)
      except Exception:  # noqa: BLE001 - already registered
        pass


_register()
