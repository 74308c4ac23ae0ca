# SQ counters of the fused block kernels (cold microbench): where do the wave cycles go?
set -u
out=gpurun_out/pmc_fused; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$out/p1 -o a -- python $R/tools/bench_fused.py f16 > $R/$out/p1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --kernel-trace --output-format csv -d $R/$out/p2 -o a -- python $R/tools/bench_fused.py f16 > $R/$out/p2.log 2>&1
cd $R
for p in p1 p2; do f=$(ls $out/$p/*counter_collection.csv $out/$p/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f | grep -E "mlp_|proj_cln|attn16" > $out/$p.txt; rm -rf $out/$p; done
tail -3 $out/p1.log
