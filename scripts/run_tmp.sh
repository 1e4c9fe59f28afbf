set -e
export UNETK_LIB=exp/libunetk_pw3.so
for g in 1024 512 256 2048 1024; do
  echo "== GRID=$g"
  UNETK_BNR_GRID=$g timeout -k 10 120 python scripts/perf_bn.py 2>&1 | grep -v "amdgpu.ids" | sed 's/apply [0-9.]* ms [0-9.]* TB.s | //; s/| bwd_apply.*//'
done
