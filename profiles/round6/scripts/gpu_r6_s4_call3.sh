# in-step launch tables of BASELINE configs 2, 5 and 4 on the round's final code (third session)
set -u
out=gpurun_out/r6t3; mkdir -p $out
for cfg in "T 32 128 4 T32" "B 32 256 4 B256" "L 128 128 5 L128"; do
  set -- $cfg
  timeout 500 python bench.py --model $1 --batch $2 --size $3 --channels $4 --no-cpu-baseline --no-parity --no-other-configs --steps 10 --launch-dump $out/l.json 2>/dev/null | tail -1 > $out/bench_$5.json
  python tools/launch_summary.py $out/l.json 60 > $out/launch_summary_$5.txt 2>&1
  rm -f $out/l.json
  python -c "import json;d=json.load(open('$out/bench_$5.json'));print('$5', d['ms_per_step'], d['phases'])" | tee -a $out/summary.txt
done
