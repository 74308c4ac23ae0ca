#!/bin/bash
set -u
out=gpurun_out/r3g; mkdir -p $out
timeout 600 python -m pytest tests/test_model_gpu.py -q -x --tb=short -k "weight_copies" 2>&1 | grep -E "^E|test_model_gpu.py:[0-9]+|passed|failed" | head -12
ABL_SET=bwd2 timeout 1500 python tools/ablate_kernels.py run bwd 2>&1 | tee $out/ablate_tail_bwd2.txt
