"""gym_solo_amd — MI355X-native batched replacement for the hot path of WPI-MMR/gym_solo.

See DESIGN.md. Import is light on purpose: the HIP engine is loaded lazily by
``gym_solo_amd.engine`` and fails loudly if ``libsolo_hip.so`` is missing.
"""
__version__ = '0.1.0'
