#!/bin/bash
set -u
out=gpurun_out/r3c; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "cln_fwd_bwd" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "presets or weight_copies or adapts or tape or ranges_are_final" 2>&1 | tail -4
bash tools/gpu_ab.sh r3c "SCOT_CLN_PARTIAL=0" "SCOT_CLN_PARTIAL=1" "SCOT_CLN_PARTIAL=0" "SCOT_CLN_PARTIAL=1" 2>&1 | cut -c1-330
