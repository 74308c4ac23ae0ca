set -u
out=gpurun_out/r6t1; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "colsum or reductions" 2>&1 | tail -4 | tee $out/pytest_colsum.txt
for i in 1 2; do
timeout 400 python bench.py --no-cpu-baseline --no-other-configs --launch-dump $out/launches.json 2>$out/bench.err | tail -1 > $out/bench_$i.json
python tools/launch_summary.py $out/launches.json 200 > $out/launch_summary_$i.txt 2>&1
rm -f $out/launches.json
python -c "import json;d=json.load(open('$out/bench_$i.json'));print(d['value'], d['ms_per_step'], d['parity']['output_rel_l2'] if isinstance(d.get('parity'),dict) else d.get('parity'))" | tee -a $out/summary.txt
done
grep -n "colsum" $out/launch_summary_1.txt | tee -a $out/summary.txt
