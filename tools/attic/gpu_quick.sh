#!/bin/bash
# One-shot kernel check on the GPU box: parity tests, bench (fused + single-step), VALU/SALU/LDS
# instruction counts per env-step.  Usage: gpurun -- bash tools/gpu_quick.sh
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
python -m pytest $R/tests -m gpu -x -q > $R/gpurun_out/t.log 2>&1 || { tail -30 $R/gpurun_out/t.log; exit 1; }
tail -1 $R/gpurun_out/t.log
for args in "" "--rollout-streams 1" "--steps-per-launch 1 --rollout-streams 1"; do
  python $R/bench.py --no-cpu-baseline $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S=%d G=%d: %.4g env-steps/s, %.2f us/step' % (d['config']['steps_per_launch'], d['config']['rollout_streams'], d['value'], d['ms_per_step']*1e3))"
done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_q
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_q -- python3 $R/tools/prof_driver.py $R/gpurun_out/pmc_q_meta.json > $R/gpurun_out/pmc_q.log 2>&1
cd $R && python tools/pmc_summary.py gpurun_out/pmc_q.json gpurun_out/pmc_q | python -c "
import sys, json
d = json.load(sys.stdin)['step']; m = json.load(open('gpurun_out/pmc_q_meta.json')); n = float(m['robots_per_launch'] * m['steps_per_launch'])
print('per env-step: VALU %.0f  SALU %.0f  LDS %.0f' % (d['SQ_INSTS_VALU'] / n, d['SQ_INSTS_SALU'] / n, d['SQ_INSTS_LDS'] / n))"
