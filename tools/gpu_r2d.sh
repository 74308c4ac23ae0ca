# Round-2 final verification + evidence run: full GPU test suite, default bench, kernel trace of the same command (per-grid table,
# per-queue step timeline), PMC traffic passes, kernel microbenchmarks.
set -u
out=gpurun_out/r2d; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $out/pytest_gpu.txt
timeout 300 python bench.py 2>$out/bench.err | tail -1 > $out/bench_default.json
python -c "import json;d=json.load(open('$out/bench_default.json'));print(d['value'], d['ms_per_step'], d['config']['parity'], d['roofline']['frac'], d['roofline']['whole_step']['frac'], d.get('cpu_baseline'))" | tee $out/summary.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o b -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity > $R/$out/prof.log 2>&1
cd $R
f=$(ls $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then
  python tools/trace_summary.py $f auto > $out/trace_by_grid.txt
  python tools/step_timeline.py $f 500 2 cpb_fwd_batched > $out/step_timeline.txt
  python tools/trace_gaps.py $f > $out/trace_gaps.txt
fi
s=$(ls $out/prof/*kernel_stats.csv $out/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$s" ] && cp $s $out/kernel_stats.csv
rm -rf $out/prof
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$out/pmc_$c -o p -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $R/$out/pmc_$c.log 2>&1
  f=$(ls $R/$out/pmc_$c/*counter_collection.csv $R/$out/pmc_$c/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f | grep -E "wgrad_group|tail_|mlp_|attn16|gemm_fast" > $R/$out/pmc_$c.txt
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f --total > $R/$out/pmc_${c}_total.txt 2>/dev/null
  rm -rf $R/$out/pmc_$c
done
cd $R
python tools/bench_kernels.py wgroup 2>&1 | grep wgrad_group > $out/micro_wgroup.txt
BK_COLD=1 python tools/bench_kernels.py attn16 2>&1 | grep attn > $out/micro_attn.txt
true
