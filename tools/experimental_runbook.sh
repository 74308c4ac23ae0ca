#!/bin/bash
# First GPU call of the next round: validate and measure the kernels that were written without GPU time (round 1).
#   gpurun --timeout 1500 -- 'bash tools/experimental_runbook.sh'
# Everything lands in gpurun_out/experimental/.  Nothing here changes defaults; flip SCOT_FUSED_MLP in engine.py only after
# (1) the gated parity tests pass, (2) the model-level presets pass with the flag on, (3) the bench is faster with it.
set -u
out=gpurun_out/experimental
mkdir -p $out
export SCOT_EXPERIMENTAL=1
echo "== 1. kernel parity (fused MLP block and projection+LN kernels vs the launches they replace)" | tee $out/summary.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "mlp_block or proj_cln" 2>&1 | tail -15 | tee -a $out/summary.txt
echo "== 2. whole-model parity with the fused kernels on (Poseidon-T/B presets, bf16 + fixtures)" | tee -a $out/summary.txt
SCOT_FUSED_MLP=1 timeout 400 python -m pytest tests/test_model_gpu.py -q -x -k "presets or bf16_vs_reference or tape" 2>&1 | tail -15 | tee -a $out/summary.txt
echo "== 3. bench A/B (Poseidon-B, batch 64, ms/step): flag off, all on, then one dimension at a time" | tee -a $out/summary.txt
b() { # label, env...
  label=$1; shift
  ms=$(env "$@" timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee $out/bench_$label.json | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$label: $ms" | tee -a $out/summary.txt
}
b off SCOT_FUSED_MLP=0
b all SCOT_FUSED_MLP=1
b c96_only SCOT_FUSED_MLP=1 SCOT_FUSED_C=96
b c192_only SCOT_FUSED_MLP=1 SCOT_FUSED_C=192
b mlp_only SCOT_FUSED_MLP=1 SCOT_FUSED_PARTS=mlp_fwd,mlp_bwd
b proj_only SCOT_FUSED_MLP=1 SCOT_FUSED_PARTS=proj_fwd,proj_bwd
echo "== 4. per-kernel times with the flag on (rocprofv3 kernel trace)" | tee -a $out/summary.txt
(cd /tmp && export TMPDIR=/tmp && SCOT_FUSED_MLP=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/prof -o fused -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OLDPWD/$out/prof.log 2>&1)
f=$(ls $out/prof/*/*kernel_stats.csv $out/prof/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && head -25 "$f" | cut -c1-200 | tee -a $out/summary.txt
