# tail_bwd<192> with launch_bounds(256, 1) where the launch has <= 256 workgroups (no scratch), and the qkv-dgrad prologue at C = 192 on top
set -u
out=gpurun_out/r6t2; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "block_tail or tail" 2>&1 | tail -3 | tee $out/pytest_tail.txt
run() { # label, env, options
  SCOT_ENGINE_OPTIONS=$3 env $2 timeout 400 python bench.py --no-cpu-baseline --no-parity --no-other-configs --steps 20 2>/dev/null | tail -1 > /tmp/_ab.json
  python - "$1" <<'PY' | tee -a $out/ab.txt
import json, sys
d = json.load(open("/tmp/_ab.json"))
print("AB", sys.argv[1], "|", round(d["ms_per_step"], 3), d["phases"])
PY
}
for rep in 1 2; do
run "minb2 (HEAD~)" SCOT_AB_TAIL_MINB2=1 ""
run "minb1" X=1 ""
run "minb1 + prologue192" X=1 "fused_qkv_dgrad=96+192"
run "minb2 + prologue192" SCOT_AB_TAIL_MINB2=1 "fused_qkv_dgrad=96+192"
done
