"""Custom solo environments (batched) — counterpart of gym_solo/envs/__init__.py."""
from gym_solo_amd.envs.solo8_base_env import Solo8BaseEnv
from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig, Solo8VanillaEnv
