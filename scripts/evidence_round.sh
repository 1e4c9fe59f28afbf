#!/bin/bash
# Counter passes over single layers (what bounds each MFMA kernel) and a repeat-run variance sample of the bench line:
#   bash scripts/evidence_round.sh <tag>      ->  gpurun_out/<tag>/{pmc_*,bench_repeat.jsonl}
set -e
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/$TAG
cd $ROOT
# conv3x3_k32_kernel<8>: 128->128 at 320x479 (epilogue-heavy) and 512->512 at 80x119 (MFMA-heavy)
bash scripts/pmc_layer.sh 3 $TAG/pmc_k32_128 > /dev/null 2>&1
bash scripts/pmc_layer.sh 9 $TAG/pmc_k32_512 > /dev/null 2>&1
# conv3x3_k32n64_kernel: 64->64 at 640x959
bash scripts/pmc_layer.sh 0 $TAG/pmc_n64 > /dev/null 2>&1
# wgrad5: 256->256 at 160x239 and 64->64 at 640x959
SCRIPT=perf_wgrad.py bash scripts/pmc_layer.sh 6 $TAG/pmc_wgrad5_256 > /dev/null 2>&1
SCRIPT=perf_wgrad.py bash scripts/pmc_layer.sh 0 $TAG/pmc_wgrad5_64 > /dev/null 2>&1
cd $ROOT
for i in 1 2 3 4 5; do
  timeout -k 10 300 python bench.py --no-cpu-baseline >> gpurun_out/$TAG/bench_repeat.jsonl 2>> gpurun_out/$TAG/bench_repeat.err
done
ls gpurun_out/$TAG
