set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_unet.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
export UNETK_LIB=exp/libunetk_pw3.so
for i in 1 2; do for v in 5 2; do
UNETK_PW_VAR=$v timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r3/ab_pw_${v}_$i.json 2> gpurun_out/r3/ab.err
python - <<PY
import json
d=json.load(open('gpurun_out/r3/ab_pw_${v}_$i.json'))
h=d['roofline_hbm']['families']
print('VAR=$v', round(d['value'], 2), round(d['ms_per_step'], 3), round(d['roofline']['frac'], 4), 'convT fwd/dgrad/wgrad', round(h['convT_fwd']['ms_per_step'],3), round(h['convT_dgrad']['ms_per_step'],3), round(h['convT_wgrad']['ms_per_step'],3))
PY
done; done
