# A/B of engine options on any bench configuration:  bash tools/ab_options.sh "<bench args>" opt1 opt2 ...   (opt = a SCOT_ENGINE_OPTIONS string)
args="$1"; shift
for o in "$@"; do
  SCOT_ENGINE_OPTIONS=$o timeout 400 python bench.py $args --no-cpu-baseline --no-parity --no-other-configs --steps 8 2>/dev/null | tail -1 > /tmp/_ab.json
  python - "$o" "$args" <<'PY'
import json, sys
d = json.load(open("/tmp/_ab.json"))
print("AB", sys.argv[2], "|", sys.argv[1], "|", round(d["ms_per_step"], 3), d["phases"])
PY
done
