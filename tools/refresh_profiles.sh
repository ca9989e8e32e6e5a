#!/bin/bash
# Regenerates the measurement artefacts of one round on the GPU box, in one gpurun call:
#   gpurun --timeout 1100 -- bash tools/refresh_profiles.sh round2_b
# Writes under gpurun_out/<tag>/; copy the summaries into profiles/ afterwards (profiles/README.md).
# rocprofv3 rules of this pool: program directly after `--`, PMC passes separate from traces.
set -e
TAG=${1:-roundX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
rm -rf $O && mkdir -p $O
cd $R
python bench.py > $O/bench_f32.jsonl 2> $O/bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_f32_driver_k20.jsonl 2>> $O/bench.err
SOLO_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_f32_rccl_forced.jsonl 2> $O/bench_rccl_forced.log
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --no-extra > $O/trace.log 2>&1
echo "trace done"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS SQ_INSTS_BRANCH SQ_INSTS_VMEM"; do
  d=$O/pmc_$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $d -- python3 $R/tools/prof_driver.py $O/prof_driver.json > $d.log 2>&1
  echo "pmc $set done"
done
cd $R
python tools/summarize_prof.py $O/trace $O/kernel_trace_bench_f32.json "bench.py --no-cpu-baseline --no-extra under rocprofv3 --kernel-trace --stats"
cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/kernel_stats_bench_f32.csv
python tools/pmc_summary.py $O/pmc_sq_f32.json $O/pmc_SQ_INSTS_VALU $O/pmc_SQ_BUSY_CYCLES $O/pmc_SQ_WAVES $O/pmc_SQ_INSTS > /dev/null
python tools/pmc_summary.py $O/pmc_hbm_f32.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > /dev/null
python tools/make_pmc_traffic.py $O > $O/pmc_traffic.log
cp profiles/pmc_traffic.json $O/pmc_traffic.json
rm -rf $O/trace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_INSTS_VALU $O/pmc_SQ_BUSY_CYCLES $O/pmc_SQ_WAVES $O/pmc_SQ_INSTS
tail -c 800 $O/bench_f32.jsonl
