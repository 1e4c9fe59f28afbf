set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_conv3x3.py -m gpu -x -q --timeout 120 --timeout-method thread -k "whole_net" 2>&1 | grep -v "^E   *+" | tail -25
