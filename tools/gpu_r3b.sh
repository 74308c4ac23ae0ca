#!/bin/bash
# Round 3, call B: new parity tests with their measured errors, the whole GPU suite, bench after the boundary/advisor batch.
set -u
out=gpurun_out/r3b; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -s -k "wgrad_group_direct" 2>&1 | grep -E "wgrad_group|passed|failed|Error|assert" | tee $out/wgrad_direct.txt | tail -25
timeout 900 python -m pytest tests/test_model_gpu.py -q -s -k "(presets and fp16) or adapts or weight_copies or L_config4" 2>&1 | grep -E "^\[|passed|failed|Error|assert|worst" | tee $out/presets_fp16.txt | tail -30
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $out/gpu_suite.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; tail -c 300 $out/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3b/bench.json'))
print('ms/step', d['ms_per_step'], 'parity', d['config']['parity'], 'ovf', d['config']['grad_overflow'])
PY
