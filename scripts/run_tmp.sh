set -e
export UNETK_LIB=exp/libunetk_pw3.so
for sm in 1024 2048; do
  echo "== NBUF=3 SMALL=$sm"
  ONLY=0 UNETK_PW_SMALL=$sm UNETK_PW_NBUF=3 timeout -k 10 120 python scripts/perf_convT.py 2>&1 | grep -v "amdgpu.ids\|wgrad"
done
