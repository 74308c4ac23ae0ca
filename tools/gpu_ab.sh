# A/B of one environment knob on the default bench: bash tools/gpu_ab.sh out_dir "ENV=a" "ENV=b" ...
set -u
out=gpurun_out/$1; shift; mkdir -p $out
for kv in "$@"; do
  label=$(echo "$kv" | tr ' =/' '___')
  env $kv timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-other-configs 2>$out/err_$label.txt | tail -1 > $out/bench_$label.json
  python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.load(open('$out/bench_$label.json'))
    t=d['config']['in_step_launches']['top']
    print("$kv", "ms/step", round(d["ms_per_step"],3), "main-ms", round(d["config"]["in_step_launches"].get("main_stream_ms_per_step",0),2), "kernel-ms", round(d['config']['in_step_launches']['kernel_ms_per_step'],2), 'launches', d['config']['in_step_launches']['launches_per_step'], {k[:12]:v['ms_per_step'] for k,v in list(t.items())[:6]}, 'ovf', d['config']['grad_overflow'])
except Exception as e:
    print('$kv', 'FAILED', e, open('$out/err_$label.txt').read()[-600:])
PY
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$out/bench_*.json'))[-1:]:
    d=json.load(open(f))
    for w in d['config']['in_step_launches'].get('wgrad_instances',[]): print(w)
PY
