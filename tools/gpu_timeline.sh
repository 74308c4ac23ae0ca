# Kernel trace of the default bench + per-queue timeline of one step (tools/step_timeline.py).  usage: bash tools/gpu_timeline.sh <outdir> [ENV=..]...
set -u
out=gpurun_out/$1; shift; mkdir -p $out
R=$PWD
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$out/prof -o b -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity > $R/$out/prof.log 2>&1
cd $R
f=$(ls $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv 2>/dev/null | head -1)
python tools/step_timeline.py $f 500 2 cpb_fwd_batched | tee $out/timeline.txt
python tools/trace_gaps.py $f | tee $out/gaps.txt
tail -1 $out/prof.log | cut -c1-300
rm -rf $out/prof
