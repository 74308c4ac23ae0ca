set -u
out=gpurun_out/r2b; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $out/pytest_gpu.txt
timeout 300 python bench.py --steps 10 --warmup 3 2>$out/bench.err | tail -1 > $out/bench_default.json
python -c "import json;d=json.load(open('$out/bench_default.json'));print(d['ms_per_step'], d['config']['parity'], d['roofline'].get('frac'), d['roofline'].get('worst_instance')); print(json.dumps(d['config']['in_step_launches'], indent=1)[:3000])" | tee $out/summary.txt
tail -3 $out/bench.err
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/prof -o fp16 -- python $OLDPWD/bench.py --no-graph --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $OLDPWD/$out/prof.log 2>&1)
f=$(ls $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python tools/trace_summary.py $f 1 copyBuffer > $out/trace_by_grid.txt
rm -rf $out/prof
true
