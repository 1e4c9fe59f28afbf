cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3/gpu_all.log 2>&1
rc=$?
tail -5 gpurun_out/r3/gpu_all.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r3/bench_e$i.json 2> gpurun_out/r3/bench_e$i.err || exit 1
python - <<PY
import json
d=json.load(open('gpurun_out/r3/bench_e$i.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['conv3x3_family']['frac'], d['roofline']['traffic'])
PY
done
