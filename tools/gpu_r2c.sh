# Round-2 verification + evidence run: full GPU test suite, default bench, kernel trace of the same command, PMC traffic passes.
set -u
out=gpurun_out/r2c; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $out/pytest_gpu.txt
timeout 300 python bench.py 2>$out/bench.err | tail -1 > $out/bench_default.json
python -c "import json;d=json.load(open('$out/bench_default.json'));print(d['value'], d['ms_per_step'], d['config']['parity'], d['roofline']['frac'], d['roofline']['whole_step']['frac'], d.get('cpu_baseline'))" | tee $out/summary.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o b -- python $R/bench.py --no-graph --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $R/$out/prof.log 2>&1
cd $R
f=$(ls $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python tools/trace_summary.py $f 11 > $out/trace_by_grid.txt
s=$(ls $out/prof/*kernel_stats.csv $out/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$s" ] && cp $s $out/kernel_stats.csv
rm -rf $out/prof
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$out/pmc_$c -o p -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $R/$out/pmc_$c.log 2>&1
  f=$(ls $R/$out/pmc_$c/*counter_collection.csv $R/$out/pmc_$c/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f | grep -E "wgrad_group|mlp_|attn16" > $R/$out/pmc_$c.txt
  rm -rf $R/$out/pmc_$c
done
true
