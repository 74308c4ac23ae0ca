#!/bin/bash
set -u
out=gpurun_out/r3d; mkdir -p $out
timeout 600 python -m pytest tests/test_model_gpu.py -q -x --tb=short -k "weight_copies" 2>&1 | tail -30
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --launch-dump $out/launches.json > $out/bench.json 2> $out/bench.err
python tools/launch_summary.py $out/launches.json 200 | grep -E "cln|steps|entry" 
