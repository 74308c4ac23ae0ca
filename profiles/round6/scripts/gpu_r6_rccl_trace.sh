set -u
out=gpurun_out/r6c6; mkdir -p $out
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$out/prof -o b -- python -m pytest $R/tests/test_model_gpu.py -m gpu -x -q -s -k "one_rank_rccl" > $R/$out/prof.log 2>&1
cd $R
f=$(ls $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv 2>/dev/null | head -1)
python tools/step_timeline.py $f 500 3 cpb_fwd_batched > $out/timeline_last.txt 2>&1
python tools/step_timeline.py $f 500 20 cpb_fwd_batched > $out/timeline_mid.txt 2>&1
python tools/step_timeline.py $f 500 40 cpb_fwd_batched > $out/timeline_early.txt 2>&1
rm -rf $out/prof
