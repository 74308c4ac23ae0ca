#!/bin/bash
# Marginal cost of each kernel family on the step (tools/whatif.py): the family's op wrapper is called twice per use ("2x", results
# unchanged for the idempotent ones, timing-only for accumulating ones) or not at all; `side` = every weight-gradient launch removed.
set -u
out=gpurun_out/r3_whatif; mkdir -p $out; rm -f $out/summary.txt
for w in none side 2xlinear_fwd 2xwindow_attn_fwd 2xcln_fwd 2xblock_tail_fwd 2xlinear_dgrad 2xwindow_attn_bwd 2xblock_tail_bwd 2xcln_bwd 2xwgrad_mlp 2xwgrad_group 2xdwconv7 2xdwconv7_wgrad 2xcpb_bwd_batched none; do
  label=$(echo $w | tr ',' '_')
  timeout 200 python tools/whatif.py $w 2>$out/err_$label.txt | tail -1 > $out/b_$label.json
  python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.load(open('$out/b_$label.json'))
    L=d['config']['in_step_launches']; ph=d['config'].get('phases') or {}
    print(f"{'$w':22s} ms/step {d['ms_per_step']:7.3f}  forward {ph.get('forward_ms', 0):6.2f}  backward {ph.get('backward_ms', 0):6.2f}  in-step kernel ms {L['kernel_ms_per_step']:6.2f}  launches {L['launches_per_step']:.0f}")
except Exception as e:
    print('$w', 'FAILED', e, open('$out/err_$label.txt').read()[-400:])
PY
done
