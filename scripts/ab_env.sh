#!/bin/bash
# same-box A/B of an environment switch:  bash scripts/ab_env.sh UNETK_FUSE_FIRST_FWD   (runs bench.py with VAR=0 / VAR=1, twice)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3
V=$1
for i in 1 2; do
for v in 0 1; do
env $V=$v timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r3/ab_${V}_${v}_$i.json 2> gpurun_out/r3/ab.err || exit 1
python - <<PY
import json
d=json.load(open('gpurun_out/r3/ab_${V}_${v}_$i.json'))
print('$V=$v', round(d['value'], 2), round(d['ms_per_step'], 3), round(d['roofline']['frac'], 4))
PY
done
done
