#!/bin/bash
set -u
out=gpurun_out/r3i; mkdir -p $out
ABL_SET=wm ABL_SRC=wgrad_mlp.hip timeout 1200 python tools/ablate_kernels.py run wm 2>&1 | tee $out/ablate_wgrad_mlp.txt
