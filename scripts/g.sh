#!/bin/bash
# gpurun helper: scripts/g.sh <log name> -> runs scripts/run_tmp.sh on the box, log under gpurun_out/r3/
mkdir -p gpurun_out/r3
bash scripts/run_tmp.sh > gpurun_out/r3/$1.log 2>&1
rc=$?
tail -${2:-80} gpurun_out/r3/$1.log
exit $rc
