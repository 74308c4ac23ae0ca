"""Average PMC counter value per dispatch, by (kernel, grid), from rocprofv3 `--pmc X --kernel-trace --output-format csv`.

usage: python tools/pmc_summary.py <counter_collection.csv>
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name).replace("unsigned short", "bf16")
    return re.sub(r"\(.*\)$", "", name)[:72]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    if "--total" in sys.argv:      # counter sum over ALL dispatches, and per training step (a step = one cpb_fwd_batched dispatch)
        steps = max(1, sum(1 for r in rows if "cpb_fwd_batched" in r["Kernel_Name"]))
        tot = defaultdict(float)
        for r in rows:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in tot.items():
            print(f"{k}: total {v:.1f} over {len(rows)} dispatches, {steps} steps -> {v / steps:.1f} per step")
        return
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows:
        k = (short(r["Kernel_Name"]), r.get("Grid_Size", ""), r["Counter_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    for (name, grid, ctr), (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:72s} grid {grid:>10s} {ctr:12s} n={n:4d} avg={v / n:14.1f}")


if __name__ == "__main__":
    main()
