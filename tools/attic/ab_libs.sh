#!/bin/bash
# Same-call A/B of two builds of the engine library on the bench workload (boxes differ by 2-3 %: only numbers from ONE
# gpurun call compare; the two libraries alternate, REPS times):
#   gpurun --timeout 900 -- bash tools/ab_libs.sh libsolo_hip_prev.so libsolo_hip.so [REPS]
# prints, per library and repetition: f32 open loop (S = 250), f32 at the driver's K = 20, one launch per step, f64 at K = 20.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
A=${1:-libsolo_hip_prev.so}; B=${2:-libsolo_hip.so}; REPS=${3:-3}
for rep in $(seq 1 $REPS); do
for lib in $A $B; do
  L=$R/gym_solo_amd/csrc/$lib
  o=$(SOLO_HIP_LIB=$L timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g' % d['value'])")
  k=$(SOLO_HIP_LIB=$L timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=20 f32 %.4g (kernel %.4f ms)  one launch per step %.4g  K=20 f64 %.4g (kernel %.4f ms)  K=20 residual opt-in %.4g' % (d['value'], d['roofline']['kernel_ms'], d.get('value_closed_loop') or 0, d.get('value_f64') or 0, d['roofline_f64']['kernel_ms'], d.get('value_residual_1e-7') or 0))")
  echo "$lib: S=250 f32 $o  $k"
done
done
