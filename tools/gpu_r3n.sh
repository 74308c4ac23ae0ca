#!/bin/bash
set -u
out=gpurun_out/r3n; mkdir -p $out
SCOT_WGRAD_GROUP_KG=2 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "wgrad_group_direct" 2>&1 | tail -2
for kg in 1 2; do SCOT_WGRAD_GROUP_KG=$kg python tools/bench_kernels.py wgroup 2>&1 | grep -i wgrad | cut -c1-200; done
bash tools/gpu_ab.sh r3n "SCOT_WGRAD_GROUP_KG=1" "SCOT_WGRAD_GROUP_KG=2" "SCOT_WGRAD_GROUP_KG=1" "SCOT_WGRAD_GROUP_KG=2" 2>&1 | cut -c1-260 | head -4
