set -u
out=gpurun_out/r2a; mkdir -p $out
timeout 600 python -m pytest tests/test_model_gpu.py -q -x -s -k "presets and fp16" 2>&1 | grep -E "^\[|passed|failed|Error|assert" | tee $out/fp16_presets.txt
for c in fp16 bf16; do
  timeout 200 python bench.py --compute $c --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > $out/bench_$c.json
  python -c "import json;d=json.load(open('$out/bench_$c.json'));print('$c', d['ms_per_step'], d['config'].get('probe_graph_ms'), d['config'].get('probe_eager_ms'), d['config'].get('eager_cpu_enqueue_ms'))" | tee -a $out/summary.txt
done
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/prof -o fp16 -- python $OLDPWD/bench.py --compute fp16 --no-graph --steps 5 --warmup 2 --no-cpu-baseline > $OLDPWD/$out/prof.log 2>&1)
f=$(ls $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python tools/trace_summary.py $f 11 > $out/trace_by_grid.txt
s=$(ls $out/prof/*kernel_stats.csv $out/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$s" ] && cp $s $out/kernel_stats.csv
rm -rf $out/prof
SCOT_STAGE_TIMING=1 timeout 200 python tools/stage_timing.py --compute fp16 2>&1 | tail -40 > $out/stage_timing.txt
true
