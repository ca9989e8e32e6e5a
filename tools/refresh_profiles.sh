#!/bin/bash
# Regenerates the measurement artefacts of one round on the GPU box:
#   gpurun --timeout 1150 -- bash tools/refresh_profiles.sh round3_a f32 f32_k20
#   gpurun --timeout 1150 -- bash tools/refresh_profiles.sh round3_a f64 f64_k20
# One CONFIG = a dtype and a launch geometry of the bench workload:
#   f32 / f64          python bench.py --dtype ...: K = 3000 steps, the engine's geometry (250 per launch, two stream slices)
#   f32_k20 / f64_k20  the driver's command, python bench.py --gpus 1 --steps 20 --warmup 5 (f64 is its default precision): ONE 4096 x 20 launch
# Per config: the bench line, a rocprofv3 --kernel-trace --stats summary of the SAME command (the step kernel's
# average duration must agree with the line's roofline.kernel_ms) and the PMC passes (separate runs of
# tools/prof_driver.py on the same geometry).  Writes under gpurun_out/<tag>/; copy the summaries into
# profiles/ afterwards (profiles/README.md).  rocprofv3 rules of this pool: program directly after `--`,
# PMC passes separate from traces.
set -e
TAG=${1:-roundX}; shift || true
CONFIGS=${@:-"f32 f32_k20 f64 f64_k20"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for CFG in $CONFIGS; do
  case $CFG in
    # (the launch geometry is the ENGINE'S in bench.py and in tools/prof_driver.py alike: round 5)
    f32)     DT=float32; BARGS="--dtype float32";                                  PENV="STEPS=2000 REPEATS=1";;
    f32_k20) DT=float32; BARGS="--dtype float32 --gpus 1 --steps 20 --warmup 5";    PENV="STEPS=20 REPEATS=40";;
    f64)     DT=float64; BARGS="--dtype float64";                                  PENV="STEPS=2000 REPEATS=1";;   # (8 launches of 250 steps on two slices: pmc_summary averages the last 6)
    f64_k20) DT=float64; BARGS="--dtype float64 --gpus 1 --steps 20 --warmup 5";    PENV="STEPS=20 REPEATS=40";;
    *) echo "unknown config $CFG"; exit 2;;
  esac
  cd $R
  # the bench line (the CPU baseline only once: it is the same sample in every line)
  if [ $CFG = f64 ] || [ $CFG = f64_k20 ]; then EXTRA=""; else EXTRA="--no-cpu-baseline"; fi
  python bench.py $BARGS $EXTRA > $O/bench_$CFG.jsonl 2> $O/bench_$CFG.err
  echo "$CFG bench done"
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/trace_$CFG
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$CFG -- python3 $R/bench.py $BARGS --no-cpu-baseline --no-extra > $O/trace_$CFG.log 2>&1
  echo "$CFG trace done"
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS SQ_INSTS_BRANCH SQ_INSTS_VMEM"; do
    d=$O/pmc_${CFG}_$(echo $set | cut -d' ' -f1)
    rm -rf $d
    export DTYPE=$DT $PENV
    rocprofv3 --pmc $set --output-format csv -d $d -- python3 $R/tools/prof_driver.py $O/prof_driver_$CFG.json > $d.log 2>&1
    echo "$CFG pmc $set done"
  done
  cd $R
  python tools/summarize_prof.py $O/trace_$CFG $O/kernel_trace_bench_$CFG.json "bench.py $BARGS --no-cpu-baseline --no-extra under rocprofv3 --kernel-trace --stats"
  cp $(find $O/trace_$CFG -name '*kernel_stats.csv' | head -1) $O/kernel_stats_bench_$CFG.csv
  python tools/pmc_summary.py $O/pmc_sq_$CFG.json $O/pmc_${CFG}_SQ_INSTS_VALU $O/pmc_${CFG}_SQ_BUSY_CYCLES $O/pmc_${CFG}_SQ_WAVES $O/pmc_${CFG}_SQ_INSTS > /dev/null
  python tools/pmc_summary.py $O/pmc_hbm_$CFG.json $O/pmc_${CFG}_FETCH_SIZE $O/pmc_${CFG}_WRITE_SIZE > /dev/null
  python tools/make_pmc_traffic.py $O $CFG > $O/pmc_traffic_$CFG.log
  rm -rf $O/trace_$CFG $O/pmc_${CFG}_FETCH_SIZE $O/pmc_${CFG}_WRITE_SIZE $O/pmc_${CFG}_SQ_INSTS_VALU $O/pmc_${CFG}_SQ_BUSY_CYCLES $O/pmc_${CFG}_SQ_WAVES $O/pmc_${CFG}_SQ_INSTS
  tail -c 600 $O/bench_$CFG.jsonl; echo
done
cp profiles/pmc_traffic.json $O/pmc_traffic.json
