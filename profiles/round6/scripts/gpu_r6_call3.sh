# Round-6 call 3: raw kernel trace of the default bench (kept for local analysis), in-step A/B of the atomic split-K policy, bench with other_configs
set -u
out=gpurun_out/r6c3; mkdir -p $out
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$out/prof -o b -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-other-configs > $R/$out/prof.log 2>&1
cd $R
f=$(ls $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python - "$f" $out/trace_small.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["start", "end", "queue", "name", "gx", "gy", "gz", "vgpr", "lds"])
qk = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
for r in rows:
    w.writerow([r["Start_Timestamp"], r["End_Timestamp"], r[qk], r["Kernel_Name"][:90], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])),
                int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])), int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_Z"])),
                r.get("VGPR_Count", ""), r.get("LDS_Block_Size", "")])
PY
rm -rf $out/prof
gzip -f $out/trace_small.csv
bash tools/gpu_ab.sh r6c3/ab "SCOT_GEMM_SPLITK=-1" "SCOT_GEMM_SPLITK=0" "SCOT_GEMM_SPLITK=-1" "SCOT_GEMM_SPLITK=0" > $out/ab_splitk.txt 2>&1
timeout 900 python bench.py 2>$out/bench.err | tail -1 > $out/bench_default.json
python -c "import json;d=json.load(open('$out/bench_default.json'));print(d['value'], d['ms_per_step'], d['parity']['output_rel_l2']); [print(o) for o in d.get('other_configs',[])]" | tee $out/summary.txt
true
