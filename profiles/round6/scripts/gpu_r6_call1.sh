# Round-6 call 1: this box's baseline of the default bench, the deep-stage GEMM table, a kernel trace with a per-step census
set -u
out=gpurun_out/r6c1; mkdir -p $out
timeout 600 python bench.py 2>$out/bench.err | tail -1 > $out/bench_default.json
python -c "import json;d=json.load(open('$out/bench_default.json'));print(d['value'], d['ms_per_step'], d['parity']['output_rel_l2'], d['phases'], d['roofline']['whole_step']['frac'])" | tee $out/summary.txt
timeout 300 python tools/bench_deep_gemm.py --model B > $out/deep_gemm_B.txt 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o b -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity > $R/$out/prof.log 2>&1
cd $R
f=$(ls $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then
  python tools/step_kernel_census.py $f > $out/step_kernel_census.txt 2>&1
  python tools/trace_summary.py $f auto > $out/trace_by_grid.txt
fi
rm -rf $out/prof
true
