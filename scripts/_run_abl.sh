cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
L=gpurun_out/r3/ls1.log
: > $L
for a in base ls1 base ls1; do
  echo "== $a" >> $L
  UNETK_LIB=$PWD/exp/libunetk_$a.so ONLY=2,3,4,5,9,13 timeout -k 10 120 python scripts/perf_conv.py >> $L 2>&1 || exit 1
done
grep -v amdgpu.ids $L
