# SQ counters of the block-tail / attention / grouped weight-gradient / NT GEMM kernels INSIDE the bench step: where do the wave cycles go?
set -u
out=gpurun_out/pmc_step; mkdir -p $out
R=$PWD
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-parity"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$out/p1 -o a -- $B > $R/$out/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --kernel-trace --output-format csv -d $R/$out/p2 -o a -- $B > $R/$out/p2.log 2>&1
cd $R
for p in p1 p2; do f=$(ls $out/$p/*counter_collection.csv $out/$p/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f | grep -E "tail_|attn16|wgrad_group_kernel|wgrad_mlp_kernel|gemm_fast_kernel<bf16, 64, 64" > $out/$p.txt; rm -rf $out/$p; done
tail -2 $out/p1.log | cut -c1-200
