"""Which kernels surround a given kernel name in a rocprofv3 kernel trace (per queue): python tools/trace_neighbours.py trace.csv NAME"""
import csv
import re
import sys
from collections import Counter, defaultdict


def short(n):
    return re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", n))[:48]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    pat = sys.argv[2]
    byq = defaultdict(list)
    for r in rows:
        byq[r["Queue_Id"]].append(r)
    ctx = Counter()
    for q, rs in byq.items():
        rs.sort(key=lambda r: int(r["Start_Timestamp"]))
        for i, r in enumerate(rs):
            if pat in r["Kernel_Name"]:
                g = r.get("Grid_Size_X", "")
                prev = short(rs[i - 1]["Kernel_Name"]) if i else "-"
                nxt = short(rs[i + 1]["Kernel_Name"]) if i + 1 < len(rs) else "-"
                ctx[(q, g, prev, nxt)] += 1
    for (q, g, prev, nxt), n in ctx.most_common(40):
        print(f"{n:6d}  queue {q} grid {g:>8s}  after [{prev}]  before [{nxt}]")


if __name__ == "__main__":
    main()
