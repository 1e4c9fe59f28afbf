#!/bin/bash
# A/B builds of the library for same-box timing:  scripts/build_variant.sh <name> [-DUNETK_...=...]...
# writes exp/libunetk_<name>.so (git-ignored; load it with UNETK_LIB=exp/libunetk_<name>.so)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/pytorch-unet_amd/csrc
OBJ=/tmp/unetk_variant_$NAME
mkdir -p $OBJ $ROOT/exp
FLAGS="--offload-arch=gfx950 -DUNETK_TUNING -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wno-unused-result"
pids=()
for f in $SRC/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $f -o $OBJ/$b.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o $ROOT/exp/libunetk_$NAME.so
echo built exp/libunetk_$NAME.so
