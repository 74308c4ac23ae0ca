# what the stand-alone layer-norm launches cost the step (VERDICT r5 "missing 4": LN as a GEMM epilogue at C = 384 / 768): tools/whatif.py,
# family skipped entirely (timing only) and run twice
set -u
out=gpurun_out/r6t6; mkdir -p $out; rm -f $out/summary.txt
for w in none cln_fwd cln_bwd cln_fwd,cln_bwd 2xcln_fwd 2xcln_bwd none; do
  label=$(echo $w | tr ',' '_')
  timeout 200 python tools/whatif.py $w --no-other-configs 2>$out/err_$label.txt | tail -1 > $out/b_$label.json
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
