#!/bin/bash
# As gpu_instr_count.sh, for single-step launches (the closed-loop step(): prologue + one step + outputs in place).
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIBS=${@:-"libsolo_hip.so"}
cd /tmp && export TMPDIR=/tmp
export STEPS=120 SPL=1 STREAMS=1
for lib in $LIBS; do
  export SOLO_HIP_LIB=$R/gym_solo_amd/csrc/$lib
  rm -rf $R/gpurun_out/pmc_s
  rocprofv3 --pmc SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/pmc_s -- python3 $R/tools/prof_driver.py $R/gpurun_out/pmc_s_meta.json > $R/gpurun_out/pmc_s.log 2>&1 || { tail -5 $R/gpurun_out/pmc_s.log; exit 1; }
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_s.json $R/gpurun_out/pmc_s | python3 -c "
import sys, json
d = json.load(sys.stdin)['step']; m = json.load(open('$R/gpurun_out/pmc_s_meta.json')); n = float(m['robots_per_launch'] * m['steps_per_launch'])
print('$lib single-step: per env-step  ALL %.0f  VALU %.0f  SALU %.0f  LDS %.0f  BRANCH %.0f  (kernel %.1f us under PMC)' % (d['SQ_INSTS'] / n, d['SQ_INSTS_VALU'] / n, d['SQ_INSTS_SALU'] / n, d['SQ_INSTS_LDS'] / n, d['SQ_INSTS_BRANCH'] / n, d['_duration_ns_under_pmc'] / 1e3))"
done
