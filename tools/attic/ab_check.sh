set -e
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 500 python -m pytest tests/test_gpu_golden.py tests/test_gpu_physics.py tests/test_gpu_env.py -m gpu -x -q > gpurun_out/t.log 2>&1 || { tail -40 gpurun_out/t.log; exit 1; }
tail -1 gpurun_out/t.log
bash tools/ab_bench.sh libsolo_hip_head.so libsolo_hip.so
bash tools/gpu_instr_count.sh libsolo_hip.so
