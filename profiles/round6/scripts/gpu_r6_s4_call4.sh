# gemm_fast direct-to-LDS ring with the MFMA fragments pipelined across the barrier (GLDS 4 / 5 / 6): kernel tests, alone (hot, graph-timed), in step
set -u
out=gpurun_out/r6t4; mkdir -p $out
for st in 4 6; do
SCOT_AB_GEMM_PIPE=$st:0:1000000 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "test_gemm or linear" 2>&1 | tail -2 | tee -a $out/pytest_gemm.txt
done
echo "== alone, hot, graph-timed (tools/probes/gemm_l2_hot_probe.py)" | tee $out/alone.txt
python tools/probes/gemm_l2_hot_probe.py 2>&1 | grep "^M=" | sed 's/^/ring3      /' | tee -a $out/alone.txt
for st in 4 5 6; do
SCOT_AB_GEMM_PIPE=$st:0:1000000 python tools/probes/gemm_l2_hot_probe.py 2>&1 | grep "^M=" | sed "s/^/pipe$st      /" | tee -a $out/alone.txt
done
run() { # label, env
  env $2 timeout 400 python bench.py --no-cpu-baseline --no-other-configs --steps 20 2>/dev/null | tail -1 > /tmp/_ab.json
  python - "$1" <<'PY' | tee -a $out/ab.txt
import json, sys
d = json.load(open("/tmp/_ab.json"))
t = d["config"]["in_step_launches"]["top"]
print("AB", sys.argv[1], "|", round(d["ms_per_step"], 3), d["phases"], "gemm NT", t.get("gemm NT (forward Linear)", {}).get("ms_per_step"), "parity", d["config"]["parity"]["output_rel_l2"])
PY
}
for rep in 1 2; do
run "ring3 (HEAD)" X=1
run "pipe4 K>=768 <=512wg" SCOT_AB_GEMM_PIPE=4:768:512
run "pipe5 K>=768 <=512wg" SCOT_AB_GEMM_PIPE=5:768:512
run "pipe6 K>=768 <=512wg" SCOT_AB_GEMM_PIPE=6:768:512
run "pipe4 all" SCOT_AB_GEMM_PIPE=4:0:100000000
run "pipe5 K>=384 <=2048wg" SCOT_AB_GEMM_PIPE=5:384:2048
done
