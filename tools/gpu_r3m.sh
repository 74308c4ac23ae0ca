#!/bin/bash
set -u
out=gpurun_out/r3m; mkdir -p $out
ABL_SET=attn ABL_KIND=bf16 ABL_SRC=attention_w16.hip timeout 1200 python tools/ablate_kernels.py run attn 2>&1 | tee $out/ablate_attn16.txt
bash tools/gpu_ab.sh r3m "SCOT_WGRAD_MLP_WGS=256" "SCOT_X=1" "SCOT_WGRAD_MLP_WGS=256" "SCOT_X=1" 2>&1 | cut -c1-200 | head -4
