"""Kernel sequence of one queue between two kernel names (the N-th occurrence of FROM up to the next TO), runs of equal names folded:
python tools/trace_window.py trace.csv FROM TO [N]"""
import csv
import re
import sys


def short(n):
    return re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", n))[:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    a, b, nth = sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 3
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if a in r["Kernel_Name"]]
    i0 = starts[min(nth, len(starts) - 1)]
    t0 = int(rows[i0]["Start_Timestamp"])
    run = None
    for r in rows[i0:]:
        name, q = short(r["Kernel_Name"]), r["Queue_Id"]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if run and run[0] == (name, q, r.get("Grid_Size_X", "")):
            run[1] += 1; run[3] = e; run[4] += e - s
        else:
            if run:
                print(f"+{(run[2] - t0) / 1e3:9.1f} us  q{run[0][1]}  x{run[1]:<4d} busy {run[4] / 1e3:8.1f} us  span {(run[3] - run[2]) / 1e3:8.1f} us  grid {run[0][2]:>8s}  {run[0][0]}")
            run = [(name, q, r.get("Grid_Size_X", "")), 1, s, e, e - s]
        if b in r["Kernel_Name"] and s > t0:
            break
    print(f"+{(run[2] - t0) / 1e3:9.1f} us  q{run[0][1]}  x{run[1]:<4d} busy {run[4] / 1e3:8.1f} us  span {(run[3] - run[2]) / 1e3:8.1f} us  grid {run[0][2]:>8s}  {run[0][0]}")


if __name__ == "__main__":
    main()
