#!/bin/bash
# Round 4's same-call A/B measurements in ONE gpurun call (boxes differ by 2-3 %: only numbers of one call compare):
#   gpurun --timeout 900 -- bash tools/round4_ab.sh > profiles/round4_ab.log
# needs gym_solo_amd/csrc/libsolo_hip_w2.so (the same sources at two waves per SIMD: make -C gym_solo_amd/csrc w2)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
f() { grep -v "amdgpu.ids"; }
echo "== f64 step kernel, two vs three waves per SIMD (same sources; -DSOLO_F64_WAVES=2), driver geometry K = 20, over batch sizes"
for lib in libsolo_hip_w2.so libsolo_hip.so; do
  SOLO_HIP_LIB=$R/gym_solo_amd/csrc/$lib timeout -k 10 250 python tools/gpu_occupancy_sweep.py float64 20 2048 3072 4096 6144 8192 12288 2>&1 | f
done
echo "== robot migration: chunk length (MIGRATE; 0 = off), f64 and f32, N = 4096, K = 20 (kernel time by HIP events)"
for dt in float64 float32; do
  for m in 0 10 5 2 1; do MIGRATE=$m timeout -k 10 250 python tools/gpu_occupancy_sweep.py $dt 20 4096 2>&1 | f | grep -v library | sed "s/^/migrate_steps $m: /"; done
done
echo "== robot migration on 250-step launches (wall clock)"
timeout -k 10 200 python tools/gpu_migrate_sweep.py float64 1000 250 2 0 125 25 2>&1 | f
timeout -k 10 200 python tools/gpu_migrate_sweep.py float64 1000 250 1 0 25 2>&1 | f
echo "== closed loop (one launch per step), f64 / f32"
for dt in float64 float32; do timeout -k 10 250 python tools/gpu_occupancy_sweep.py $dt 1 4096 2>&1 | f | grep -v library; done
echo "== longest-first launch order, host-side upper bound (f64, K = 20 and K = 1)"
timeout -k 10 250 python tools/gpu_lpt_probe.py float64 20 2>&1 | f
timeout -k 10 250 python tools/gpu_lpt_probe.py float64 1 2>&1 | f
