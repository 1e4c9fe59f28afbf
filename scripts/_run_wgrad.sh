cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 600 python -u -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -m gpu -x -q -k "wgrad" --timeout 300 --timeout-method thread > gpurun_out/r3/w5_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r3/w5_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python scripts/perf_wgrad.py > gpurun_out/r3/wgrad_w5.log 2>&1
cat gpurun_out/r3/wgrad_w5.log
