"""The README's quick-start snippet, executed as written (with a smaller batch)."""
import os
import re

import pytest

pytestmark = pytest.mark.gpu


def test_readme_quick_start_runs():
  import torch
  if not torch.cuda.is_available():
    pytest.fail('GPU tests need a visible MI355X')
  text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'README.md')).read()
  code = re.search(r'## Quick start\s+```python\n(.*?)```', text, re.S).group(1)
  code = code.replace('4096', '128').replace('1000, 128, 12', '300, 128, 12')   # (TimeBasedTermination(1000) stays)
  scope = {}
  exec(compile(code, 'README.md quick start', 'exec'), scope)   # noqa: S102 - our own documentation
  assert scope['o'].shape == (128, 21) and scope['r'].shape == (128,) and scope['done'].dtype == torch.bool
  assert scope['obs_k'].shape == (300, 128, 21) and scope['done_k'].shape == (300, 128)
  assert bool(torch.isfinite(scope['obs_k']).all())
  o, r, term, trunc, info = scope['venv'].step(torch.zeros(128, 12, device='cuda'))
  assert o.shape == (128, 21)
  scope['env']._close()
