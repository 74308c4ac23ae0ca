#!/bin/bash
set -u
out=gpurun_out/r3l; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "lean_forms" 2>&1 | tail -2
for w in 256 384 512; do SCOT_WGRAD_MLP_WGS=$w python tools/bench_wgrad_mlp.py; done
bash tools/gpu_ab.sh r3l "SCOT_LEAN_TAIL=0" "SCOT_LEAN_TAIL=1" "SCOT_LEAN_TAIL=1 SCOT_WGRAD_MLP_WGS=512" "SCOT_LEAN_TAIL=0" "SCOT_LEAN_TAIL=1" 2>&1 | cut -c1-300 | head -5
