# Round-6 call 7: the full GPU suite, the default bench (with other_configs), the six-models-in-a-row probe against the new stream measurement
set -u
out=gpurun_out/r6c7; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $out/pytest_gpu.txt
timeout 900 python bench.py 2>$out/bench.err | tail -1 > $out/bench_default.json
python -c "import json;d=json.load(open('$out/bench_default.json'));print(d['value'], d['ms_per_step'], d['parity']['output_rel_l2'], d['roofline']['attainable']['ms_per_step']); [print(o.get('workload'), o.get('ms_per_step'), o.get('parity_output_rel_l2')) for o in d.get('other_configs',[])]" | tee $out/summary.txt
timeout 600 python tools/probe_second_model4.py 2>&1 | grep "^model" | tee $out/six_models.txt
true
