"""Type aliases — counterpart of gym_solo/solo_types.py:5-12 (batched: tensors, not scalars)."""
from typing import Any

# A batch of state observations, [N, D]
obs = Any

# A batch of rewards after a step, [N]
reward = Any

# Return value for "no-op" functions when monkey-patching
no_op = 'NO_OP'
