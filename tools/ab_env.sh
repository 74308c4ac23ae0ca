# A/B of environment settings on any bench configuration:  bash tools/ab_env.sh "<bench args>" "ENV=a" "ENV=b" ...
args="$1"; shift
for kv in "$@"; do
  env $kv timeout 400 python bench.py $args --no-cpu-baseline --no-parity --no-other-configs --steps 8 2>/dev/null | tail -1 > /tmp/_ab.json
  python - "$kv" "$args" <<'PY'
import json, sys
d = json.load(open("/tmp/_ab.json"))
print("AB", sys.argv[2], "|", sys.argv[1], "|", round(d["ms_per_step"], 3), d["phases"])
PY
done
