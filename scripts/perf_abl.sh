for D in 0 1 2 4 5 3; do echo "DBG=$D"; UNETK_CONV_DBG=$D STATS=${STATS:-1} timeout -k 10 120 python -u scripts/perf_conv.py 2>&1 | grep -E "64->64|512->512|1024->1024"; done
