#!/bin/bash
set -u
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "dwconv" 2>&1 | tail -2
BK_COLD=1 python tools/bench_kernels.py 2>&1 | grep -i "dwconv"
bash tools/gpu_ab.sh r3z "X=1" "X=2" "X=3" 2>&1 | cut -c1-120 | head -3
