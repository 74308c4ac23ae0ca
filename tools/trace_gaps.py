"""Idle gaps between consecutive kernels of each HIP queue in a rocprofv3 kernel_trace.csv (launch gaps on the dependent chain).
usage: python tools/trace_gaps.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
    per = defaultdict(list)
    for r in rows:
        per[r[qkey]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
    for q, ks in sorted(per.items(), key=lambda kv: -len(kv[1])):
        ks.sort()
        gaps = [max(0, ks[i][0] - ks[i - 1][1]) for i in range(1, len(ks))]
        small = [g for g in gaps if g < 50_000]          # ignore host-side pauses between steps
        busy = sum(e - s for s, e, _ in ks)
        if not small:
            continue
        small.sort()
        print(f"queue {q}: {len(ks)} kernels, busy {busy / 1e6:.1f} ms, gaps<50us: n={len(small)} sum {sum(small) / 1e6:.2f} ms "
              f"median {small[len(small) // 2] / 1e3:.1f} us p90 {small[int(len(small) * 0.9)] / 1e3:.1f} us; gaps>=50us: {len(gaps) - len(small)}")


if __name__ == "__main__":
    main()
