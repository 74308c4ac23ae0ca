#!/bin/bash
set -u
out=gpurun_out/r3h; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "lean_forms or block_tail or cln_fwd_bwd" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "presets or L_config4 or tape or trainer or training_run" 2>&1 | tail -4
bash tools/gpu_ab.sh r3h "SCOT_LEAN_TAIL=0" "SCOT_LEAN_TAIL=1" "SCOT_LEAN_TAIL=0" "SCOT_LEAN_TAIL=1" 2>&1 | cut -c1-320
SCOT_LEAN_TAIL=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --launch-dump $out/launches.json > $out/bench.json 2> $out/bench.err
python tools/launch_summary.py $out/launches.json 40 | cut -c1-150
python -c "
import json; d=json.load(open('$out/bench.json')); print('lean: ms', d['ms_per_step'], d['config']['parity'])"
