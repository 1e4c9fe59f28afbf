cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
L=gpurun_out/r3/tpbmax.log
: > $L
export UNETK_LIB=$PWD/exp/libunetk_tpb.so
for m in 16 96 16 96; do
  echo "== TPB_MAX=$m" >> $L
  UNETK_CONV_TPB_MAX=$m ONLY=0,1,2,3,13 timeout -k 10 120 python scripts/perf_conv.py >> $L 2>&1 || exit 1
  UNETK_CONV_TPB_MAX=$m timeout -k 10 120 python scripts/perf_convT.py 2>&1 | grep -v wgrad >> $L || exit 1
done
grep -v amdgpu.ids $L
