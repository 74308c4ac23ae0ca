set -u
out=gpurun_out/r5e; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "wide" > $out/pytest_gemm.txt 2>&1; tail -3 $out/pytest_gemm.txt
for m in B L B256; do timeout 900 python tools/bench_deep_gemm.py --model $m --json $out/deep_gemm_$m.json > $out/deep_gemm_$m.txt 2>&1; cat $out/deep_gemm_$m.txt; done
bash tools/gpu_ab.sh r5e "SCOT_GEMM_WIDE=0" "SCOT_GEMM_WIDE=1" "SCOT_GEMM_WIDE=0" "SCOT_GEMM_WIDE=1"
for w in 0 1; do
SCOT_GEMM_WIDE=$w timeout 300 python bench.py --model L --batch 128 --channels 5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('L wide=$w', d['ms_per_step'], d['config']['parity']['output_rel_l2'] if d['config']['parity'] else None, {k:v['ms_per_step'] for k,v in list(d['roofline']['families'].items())[:5]})"
SCOT_GEMM_WIDE=$w timeout 300 python bench.py --model B --batch 32 --size 256 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B256 wide=$w', d['ms_per_step'], d['config']['parity']['output_rel_l2'] if d['config']['parity'] else None, {k:v['ms_per_step'] for k,v in list(d['roofline']['families'].items())[:5]})"
done
