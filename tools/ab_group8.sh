#!/bin/bash
# Same-call A/B of the EXPERIMENT build `make -C gym_solo_amd/csrc group8` (8 robots per workgroup, their leg dynamics
# computed by ONE wave: solo_step_kernel_g8.h) against the product library: the round-2 review's "measure, don't argue,
# the 8-way redundant dynamics".  Boxes differ by 2-3 %: only numbers from ONE gpurun call compare.
#   gpurun --timeout 900 -- bash tools/ab_group8.sh > profiles/round3_group8_ab.log
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
for lib in libsolo_hip.so libsolo_hip_group8.so; do
  for n in 4096 8192; do
    for args in "" "--steps 20 --warmup 5"; do
      SOLO_HIP_LIB=$R/gym_solo_amd/csrc/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --envs-per-gpu $n $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib N=$n [$args]: f32 %.4g env-steps/s (kernel %.4g ms), one launch per step %.4g, f64 %.4g (kernel %.4g ms)' % (d['value'], d['roofline']['kernel_ms'], d.get('value_closed_loop') or 0, d.get('value_f64') or 0, d['roofline_f64']['kernel_ms']))"
    done
  done
done
done
