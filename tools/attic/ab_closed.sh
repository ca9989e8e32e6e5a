#!/bin/bash
# Same-call A/B of step-kernel builds on the closed-loop figure only (one repeat per build per round).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do for lib in $@; do
SOLO_HIP_LIB=$R/gym_solo_amd/csrc/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib: K=20 %.4g closed %.4g' % (d['value'], d.get('value_closed_loop') or 0))"
done; done
