#!/bin/bash
# Everything profiles/ holds for one round, on the GPU box:  bash scripts/profile_round.sh r02
# (then, back here:  python scripts/profile_collect.py r02)
set -e
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
# 1. the driver's command
timeout -k 10 400 $B > $OUT/bench_default.json 2> $OUT/bench_default.err
# 2. kernel trace of the same command (fewer steps)
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $B --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
# 3. counters, one pass each (8 SQ / 4 TCC slots; FETCH_SIZE takes 3, WRITE_SIZE 2)
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $OUT/pmc_mfma -o m -- $B --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_mfma.json 2> $OUT/pmc_mfma.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f --output-format csv -- $B --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w --output-format csv -- $B --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.err
# 4. the other BASELINE.json configurations, same harness
for c in infer959_bilinear train1918 train512_c21; do
  timeout -k 10 400 $B --config $c --steps 5 --warmup 2 >> $OUT/other_configs.jsonl 2>> $OUT/other_configs.err
done
ls $OUT
