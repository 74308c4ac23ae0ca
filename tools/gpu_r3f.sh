#!/bin/bash
set -u
out=gpurun_out/r3f; mkdir -p $out
timeout 600 python -m pytest tests/test_model_gpu.py -q -x --tb=short -k "weight_copies" 2>&1 | grep -E "^E|test_model_gpu.py:[0-9]+|passed|failed" | head -12
timeout 1500 python tools/ablate_kernels.py run fwd bwd 2>&1 | tee $out/ablate_tail.txt
