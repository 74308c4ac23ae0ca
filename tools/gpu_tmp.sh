#!/bin/bash
set -u
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "dwconv" 2>&1 | tail -2
for w in 1 384; do echo "WGS=$w"; SCOT_DWCONV_WGRAD_WGS=$w BK_COLD=1 python tools/bench_kernels.py dwconv 2>&1 | grep -i "wgrad"; done
BK_COLD=1 python tools/bench_kernels.py 2>&1 | grep -i "dwconv"
bash tools/gpu_ab.sh r3y "SCOT_DWCONV_WGRAD_WGS=1" "SCOT_DWCONV_WGRAD_WGS=384" "SCOT_DWCONV_WGRAD_WGS=1" "SCOT_DWCONV_WGRAD_WGS=384" 2>&1 | cut -c1-120 | head -4
