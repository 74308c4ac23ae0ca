#!/bin/bash
set -u
out=gpurun_out/r3e; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "cln_fwd_bwd or gemm_layouts or gemm_epilogues" 2>&1 | tail -3
SCOT_GEMM_KG_NKT=6 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_layouts or gemm_epilogues or transpose_cast" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_model_gpu.py -q -x -k "weight_copies" 2>&1 | tail -3
bash tools/gpu_ab.sh r3e "SCOT_GEMM_KG_NKT=0" "SCOT_GEMM_KG_NKT=24" "SCOT_GEMM_KG_NKT=12" "SCOT_GEMM_KG_NKT=6" "SCOT_GEMM_KG_NKT=0" "SCOT_GEMM_KG_NKT=24" 2>&1 | cut -c1-300
SCOT_GEMM_KG_NKT=12 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --launch-dump $out/launches.json > $out/bench.json 2> $out/bench.err
python tools/launch_summary.py $out/launches.json 200 | grep -E "cln_bwd|gemm  |steps|entry" | head -40
