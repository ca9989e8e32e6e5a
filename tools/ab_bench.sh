set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_golden.py tests/test_gpu_physics.py -m gpu -x -q > gpurun_out/t.log 2>&1 || { tail -40 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
for rep in 1 2; do
for lib in libsolo_hip_head.so libsolo_hip.so; do
  for args in "" "--steps 20 --warmup 5"; do
    SOLO_HIP_LIB=$R/gym_solo_amd/csrc/$lib timeout -k 10 200 python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib [$args]: %.4g env-steps/s, closed %.4g, f64 %.4g' % (d['value'], d.get('value_closed_loop') or 0, d.get('value_f64') or 0))"
  done
done
done
