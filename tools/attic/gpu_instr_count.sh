#!/bin/bash
# Dynamic instruction counts per env-step of step-kernel builds (SQ_INSTS* counters, one rocprofv3 pass per
# build): a wave issues one instruction every ~7 cycles whatever it is, so THIS is the number the kernel's
# time follows - and unlike a timing it does not depend on the box.
# Usage: gpurun -- bash tools/gpu_instr_count.sh libA.so libB.so ...   (files under gym_solo_amd/csrc)
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIBS=${@:-"libsolo_hip.so"}
cd /tmp && export TMPDIR=/tmp
export STEPS=1000
for lib in $LIBS; do
  export SOLO_HIP_LIB=$R/gym_solo_amd/csrc/$lib
  rm -rf $R/gpurun_out/pmc_i
  rocprofv3 --pmc SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/pmc_i -- python3 $R/tools/prof_driver.py $R/gpurun_out/pmc_i_meta.json > $R/gpurun_out/pmc_i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_i.log; exit 1; }
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_i.json $R/gpurun_out/pmc_i | python3 -c "
import sys, json
d = json.load(sys.stdin)['step']; m = json.load(open('$R/gpurun_out/pmc_i_meta.json')); n = float(m['robots_per_launch'] * m['steps_per_launch'])
print('$lib: per env-step  ALL %.0f  VALU %.0f  SALU %.0f  LDS %.0f  BRANCH %.0f  (kernel %.3f ms under PMC)' % (d['SQ_INSTS'] / n, d['SQ_INSTS_VALU'] / n, d['SQ_INSTS_SALU'] / n, d['SQ_INSTS_LDS'] / n, d['SQ_INSTS_BRANCH'] / n, d['_duration_ns_under_pmc'] / 1e6))"
done
