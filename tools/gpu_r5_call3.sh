set -u
export SCOT_GEMM_WIDE=0
bash tools/gpu_ab.sh r5c "SCOT_RECOMPUTE_TAIL=0" "SCOT_RECOMPUTE_TAIL=1" "SCOT_RECOMPUTE_TAIL=0" "SCOT_RECOMPUTE_TAIL=1"
timeout 900 python tools/probe_graph.py --out gpurun_out/r5c/graph 2>&1 | tail -12
ls -la gpurun_out/r5c/graph
