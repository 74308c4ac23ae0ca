#!/bin/bash
set -u
out=gpurun_out/r3k; mkdir -p $out
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "presets or tape or ranges_are_final" 2>&1 | tail -2
bash tools/gpu_ab.sh r3k "SCOT_LEAN_TAIL=0 SCOT_CLN_PARTIAL=0" "SCOT_LEAN_TAIL=0 SCOT_CLN_PARTIAL=1" "SCOT_LEAN_TAIL=1 SCOT_CLN_PARTIAL=1" "SCOT_LEAN_TAIL=1 SCOT_CLN_PARTIAL=0" "SCOT_LEAN_TAIL=0 SCOT_CLN_PARTIAL=0" "SCOT_LEAN_TAIL=1 SCOT_CLN_PARTIAL=1" 2>&1 | cut -c1-330
