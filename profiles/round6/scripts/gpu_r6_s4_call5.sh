# gemm_fast four-stage ring, fragments pipelined across the barrier with asm reads / waits / MFMAs: kernel tests, alone (hot, graph-timed), in step
set -u
out=gpurun_out/r6t5; mkdir -p $out
SCOT_AB_GEMM_PIPE=0:1000000 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "test_gemm or linear" 2>&1 | grep -E "passed|failed" | tee $out/pytest_gemm.txt
echo "== alone, hot, graph-timed (tools/probes/gemm_l2_hot_probe.py)" | tee $out/alone.txt
python tools/probes/gemm_l2_hot_probe.py 2>&1 | grep "^M=" | sed 's/^/ring3      /' | tee -a $out/alone.txt
SCOT_AB_GEMM_PIPE=0:1000000 python tools/probes/gemm_l2_hot_probe.py 2>&1 | grep "^M=" | sed "s/^/pipe4asm   /" | tee -a $out/alone.txt
run() { # label, env
  env $2 timeout 400 python bench.py --no-cpu-baseline --no-other-configs --steps 20 2>/dev/null | tail -1 > /tmp/_ab.json
  python - "$1" <<'PY' | tee -a $out/ab.txt
import json, sys
d = json.load(open("/tmp/_ab.json"))
t = d["config"]["in_step_launches"]["top"]
print("AB", sys.argv[1], "|", round(d["ms_per_step"], 3), "fwd %.3f bwd %.3f" % (d["phases"]["forward_ms"], d["phases"]["backward_ms"]), "gemm NT", t.get("gemm NT (forward Linear)", {}).get("ms_per_step"), "parity", d["config"]["parity"]["output_rel_l2"])
PY
}
for rep in 1 2 3; do
run "ring3 (HEAD)" X=1
run "pipe4asm K>=768 <=512wg" SCOT_AB_GEMM_PIPE=768:512
run "pipe4asm K>=384 <=512wg" SCOT_AB_GEMM_PIPE=384:512
run "pipe4asm K>=192 <=2048wg" SCOT_AB_GEMM_PIPE=192:2048
done
