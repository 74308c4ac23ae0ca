#!/bin/bash
set -u
out=gpurun_out/r3j; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "lean_forms or block_tail" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "presets or tape" 2>&1 | tail -2
bash tools/gpu_ab.sh r3j "SCOT_LEAN_TAIL=0" "SCOT_LEAN_TAIL=1" "SCOT_LEAN_TAIL=1 SCOT_WGRAD_MLP_WGS=512" "SCOT_LEAN_TAIL=1 SCOT_LEAN_DACT=recompute" "SCOT_LEAN_TAIL=0" "SCOT_LEAN_TAIL=1" 2>&1 | cut -c1-330
