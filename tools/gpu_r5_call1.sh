# round 5, call 1: gemm_wide on the GPU — parity, per-shape sweep, in-step A/B
set -u
out=gpurun_out/r5a; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "wide or gemm" > $out/pytest_gemm.txt 2>&1; tail -3 $out/pytest_gemm.txt
timeout 400 python tools/bench_deep_gemm.py --json $out/deep_gemm_B.json > $out/deep_gemm_B.txt 2>&1; cat $out/deep_gemm_B.txt
bash tools/gpu_ab.sh r5a "SCOT_GEMM_WIDE=0" "SCOT_GEMM_WIDE=1" "SCOT_GEMM_WIDE=0" "SCOT_GEMM_WIDE=1"
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>$out/bench_full_err.txt | tail -1 > $out/bench_full_wide1.json
python - <<PY
import json
d=json.load(open('$out/bench_full_wide1.json'))
print('full bench: ms', d['ms_per_step'], 'parity', d['config']['parity'], 'phases', d['config']['phases'])
print({k:(v['ms_per_step'],v['launches_per_step']) for k,v in d['roofline']['families'].items()})
PY
