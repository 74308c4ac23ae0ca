# Round-6 evidence run (what profiles/round6/ is copied from): default bench (+ per-launch dump), kernel trace of the same command
# (kernel stats, per-grid table, per-queue step timeline), FETCH_SIZE / WRITE_SIZE PMC passes, bare vs torchrun-1, the other configs.
#   usage: bash tools/gpu_r6_evidence.sh [tests]   ("tests": run the full GPU suite first)
set -u
out=gpurun_out/r6ev; mkdir -p $out
if [ "${1:-}" = tests ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $out/pytest_gpu.txt
fi
timeout 400 python bench.py --launch-dump $out/launches.json 2>$out/bench.err | tail -1 > $out/bench_default.json
python tools/launch_summary.py $out/launches.json 80 > $out/launch_summary.txt 2>&1
rm -f $out/launches.json
python -c "import json;d=json.load(open('$out/bench_default.json'));print(d['value'], d['ms_per_step'], d['parity'], d['roofline']['frac'], d['roofline']['kernel'][:40], d['roofline']['whole_step']['frac'], d['roofline'].get('hbm'), d.get('cpu_baseline'))" | tee $out/summary.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o b -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-other-configs > $R/$out/prof.log 2>&1
cd $R
f=$(ls $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then
  python tools/trace_summary.py $f auto > $out/trace_by_grid.txt
  python tools/step_timeline.py $f 500 2 cpb_fwd_batched > $out/step_timeline.txt
  python tools/trace_gaps.py $f > $out/trace_gaps.txt
  python tools/step_kernel_census.py $f > $out/step_kernel_census.txt
fi
s=$(ls $out/prof/*kernel_stats.csv $out/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$s" ] && cp $s $out/kernel_stats.csv
rm -rf $out/prof
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$out/pmc_$c -o p -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-other-configs > $R/$out/pmc_$c.log 2>&1
  f=$(ls $R/$out/pmc_$c/*counter_collection.csv $R/$out/pmc_$c/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f | head -120 > $R/$out/pmc_$c.txt
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f --total > $R/$out/pmc_${c}_total.txt 2>/dev/null
  rm -rf $R/$out/pmc_$c
done
cd $R
python tools/make_pmc_traffic.py $out/pmc_FETCH_SIZE.txt $out/pmc_WRITE_SIZE.txt $out/pmc_FETCH_SIZE_total.txt $out/pmc_WRITE_SIZE_total.txt > $out/pmc_traffic.json 2>$out/pmc_traffic.err
# process-group overhead: the same bench under a 1-rank RCCL group (exposed exchange, RCCL's own log parsed into the line)
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline --no-parity --no-other-configs 2>$out/torchrun1.err | tail -1 > $out/bench_torchrun1.json
timeout 300 python bench.py --no-cpu-baseline --no-parity --no-other-configs 2>/dev/null | tail -1 > $out/bench_bare.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --rccl-channels 16 --no-cpu-baseline --no-parity --no-other-configs 2>/dev/null | tail -1 > $out/bench_torchrun1_ch16.json
# the non-headline configurations are part of the default line (`other_configs`); here: the bfloat16 build and the 128 x 128-tile GEMMs switched off
timeout 400 python bench.py --compute bf16 --no-cpu-baseline --no-other-configs --steps 10 2>/dev/null | tail -1 > $out/bench_bf16.json
SCOT_GEMM_WIDE=0 timeout 400 python bench.py --model L --batch 128 --channels 5 --no-cpu-baseline --no-parity --steps 10 2>/dev/null | tail -1 > $out/bench_L_wide0.json
python tools/bench_inference.py 64 2>&1 | grep inference > $out/inference.txt
true
