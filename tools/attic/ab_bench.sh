#!/bin/bash
# Same-call A/B of step-kernel builds (boxes differ by 2-3 %: only numbers from ONE gpurun call compare).
# Usage: gpurun -- bash tools/ab_bench.sh libA.so libB.so ...   (files under gym_solo_amd/csrc; default: head vs product)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
LIBS=${@:-"libsolo_hip_head.so libsolo_hip.so"}
for rep in 1 2; do
for lib in $LIBS; do
  for args in "" "--steps 20 --warmup 5"; do
    SOLO_HIP_LIB=$R/gym_solo_amd/csrc/$lib timeout -k 10 200 python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib [$args]: %.4g env-steps/s, closed %.4g, f64 %.4g' % (d['value'], d.get('value_closed_loop') or 0, d.get('value_f64') or 0))"
  done
done
done
