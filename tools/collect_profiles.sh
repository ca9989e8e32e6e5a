#!/bin/bash
# Copies the judged summaries of one refresh_profiles.sh run from gpurun_out/<tag>/ into profiles/<tag>_* and installs its
# pmc_traffic.json as profiles/pmc_traffic.json (run it AFTER the last edit of the kernel sources: bench.py drops a
# counter profile whose kernel_source_hash differs from the tree's).   bash tools/collect_profiles.sh round4_d
set -e
TAG=$1; R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/$TAG
for f in $O/bench_*.jsonl $O/kernel_stats_bench_*.csv $O/kernel_trace_bench_*.json $O/pmc_hbm_*.json $O/pmc_sq_*.json $O/prof_driver_*.json; do
  cp $f $R/profiles/${TAG}_$(basename $f)
done
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
python3 - <<PY
import json, sys
sys.path.insert(0, "$R")
from gym_solo_amd.build_info import kernel_source_hash
t = json.load(open("$R/profiles/pmc_traffic.json"))
now = kernel_source_hash()
for k, v in t.items():
  print(k, v.get("kernel_source_hash"), "OK" if v.get("kernel_source_hash") == now else "STALE (tree: %s)" % now)
PY
