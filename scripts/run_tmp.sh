set -e
export UNETK_LIB=exp/libunetk_pw3.so
for d in 8 0 8 0; do
  echo "== DBG=$d (8: no relaxed wait)"
  UNETK_CONV_DBG=$d timeout -k 10 120 python scripts/perf_convT.py 2>&1 | grep -v "amdgpu.ids\|wgrad"
done
timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_conv3x3.py -m gpu -x -q -k "convT or up_convT or whole_net" --timeout 120 --timeout-method thread 2>&1 | tail -3
