# Round-6 call 4: the new GPU tests, in-step A/B of the lazy zero-grad, the B@256 puzzle (other_configs 46 ms vs 30 ms stand-alone)
set -u
out=gpurun_out/r6c4; mkdir -p $out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "store_and_scaled or segments or split_k_atomic or lazy or grad_accumulation or overlapped_gradient or grad_ranges or wgrad_group or lean" 2>&1 | tail -8 | tee $out/pytest_new.txt
bash tools/gpu_ab.sh r6c4/ab "SCOT_ENGINE_OPTIONS=lazy_grads=0" "SCOT_ENGINE_OPTIONS=lazy_grads=1" "SCOT_ENGINE_OPTIONS=lazy_grads=0" "SCOT_ENGINE_OPTIONS=lazy_grads=1" > $out/ab_lazy.txt 2>&1
timeout 400 python bench.py --model B --size 256 --batch 32 --no-cpu-baseline --no-parity --steps 10 2>/dev/null | tail -1 > $out/bench_B256.json
python -c "import json;d=json.load(open('$out/bench_B256.json'));print('B256 standalone', d['ms_per_step'], d['config'].get('probe_graph_ms'), d['config'].get('probe_eager_ms'), d['phases'])" | tee $out/b256.txt
true
