#!/bin/bash
# Round 3, call A: baseline + per-launch dump + timing-only what-ifs (upper bounds of the planned changes) + two GEMM knobs.
set -u
out=gpurun_out/r3a; mkdir -p $out
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --launch-dump $out/launches.json > $out/bench_base.json 2> $out/bench_base.err
tail -c 600 $out/bench_base.err
python tools/launch_summary.py $out/launches.json 70 > $out/launch_summary.txt 2>&1
head -75 $out/launch_summary.txt
bash tools/gpu_ab.sh r3a "SCOT_WHATIF=" "SCOT_WHATIF=noact" "SCOT_WHATIF=noact,nomlpwgrad" "SCOT_WHATIF=nocast" "SCOT_GEMM_NT_SPLIT=1" "SCOT_GEMM_TILE_NT=3"
