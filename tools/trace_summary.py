"""Aggregate a rocprofv3 kernel_trace.csv by (kernel, grid): calls, average and total time, plus busy/idle totals.

usage: python tools/trace_summary.py <kernel_trace.csv> [steps | auto] > summary.txt
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("unsigned short", "bf16")
    name = re.sub(r"\(.*\)$", "", name)
    return name[:70]


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    if len(sys.argv) > 2 and sys.argv[2] == "auto":     # one cpb_fwd_batched dispatch per training forward
        steps = max(1, sum(1 for r in rows if "cpb_fwd_batched" in r["Kernel_Name"]))
    else:
        steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    agg = defaultdict(lambda: [0, 0.0])
    spans = []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        grid = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])),
                int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_Z"])))
        k = (short(r["Kernel_Name"]), grid)
        agg[k][0] += 1
        agg[k][1] += (e - s)
        spans.append((s, e))
    if len(sys.argv) > 3:   # show what runs around a kernel: python trace_summary.py trace.csv steps <substring>
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        hits = [i for i, r in enumerate(rows) if sys.argv[3] in r["Kernel_Name"]]
        for i in hits[len(hits) // 2: len(hits) // 2 + 12]:
            print(" | ".join(short(rows[j]["Kernel_Name"])[:40] for j in range(max(0, i - 2), min(len(rows), i + 2))))
    spans.sort()
    busy, cur_s, cur_e = 0, None, None
    for s, e in spans:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    total = sum(v[1] for v in agg.values())
    wall = spans[-1][1] - spans[0][0]
    print(f"dispatches {len(rows)}  sum-of-kernels {total/1e6:.2f} ms  union-busy {busy/1e6:.2f} ms  first-to-last {wall/1e6:.2f} ms  (/{steps} steps: "
          f"{total/1e6/steps:.2f} / {busy/1e6/steps:.2f} ms)")
    print(f"{'kernel':70s} {'grid':>16s} {'calls/step':>10s} {'avg us':>9s} {'ms/step':>8s} {'%':>6s}")
    for (name, grid), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:160]:
        print(f"{name:70s} {str(grid):>16s} {n/steps:10.1f} {t/n/1e3:9.1f} {t/1e6/steps:8.3f} {100*t/total:6.2f}")


if __name__ == "__main__":
    main()
