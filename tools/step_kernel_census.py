"""Kernel census of the TIMED steps of a rocprofv3 kernel_trace.csv: dispatches per step by kernel name, counted inside real step windows
(from one forward's first kernel to the next one's) instead of dividing the whole trace by a step count — bench.py's process also runs
warm-up, parity, per-launch-timed and phase-timed steps, whose helper dispatches (event-pair replays, clones of outputs) would otherwise
be charged to the timed step.
usage: python tools/step_kernel_census.py <kernel_trace.csv> [marker=cpb_fwd_batched] [pattern=rocclr|Fill|elementwise]"""
import csv
import re
import sys
from collections import Counter


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    marker = sys.argv[2] if len(sys.argv) > 2 else "cpb_fwd_batched"
    pat = re.compile(sys.argv[3] if len(sys.argv) > 3 else r"rocclr|Fill|elementwise|at::")
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                 int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in rows)
    starts = [s for s, e, n, g in ks if marker in n]
    if len(starts) < 3:
        print("not enough steps")
        return
    wins = list(zip(starts[:-1], starts[1:]))
    # the timed region = the longest run of consecutive windows of near-identical length (replayed steps back to back)
    lens = [b - a for a, b in wins]
    med = sorted(lens)[len(lens) // 2]
    print(f"{len(wins)} step windows in the trace; median window {med / 1e6:.3f} ms")
    total = Counter()
    print("window  ms      kernels  " + "matching dispatches (name x grid: count)")
    for i, (a, b) in enumerate(wins):
        inside = [(n, g) for s, e, n, g in ks if a <= s < b]
        c = Counter((re.sub(r"\(.*$", "", n.replace("void ", ""))[:44], g) for n, g in inside if pat.search(n))
        tag = "steady" if abs((b - a) - med) < 0.15 * med else "other "
        print(f"{i:4d}  {(b - a) / 1e6:7.3f}  {len(inside):6d}  {tag}  " + "; ".join(f"{k[0]} x{k[1]}: {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:8]))
        if tag == "steady":
            total.update({k: v for k, v in c.items()})
            total["__n__"] += 1
    n = max(1, total.pop("__n__", 1))
    print(f"per steady step (mean over {n} windows):")
    for k, v in sorted(total.items(), key=lambda kv: -kv[1]):
        print(f"  {v / n:7.2f}  {k[0]} grid {k[1]}")


if __name__ == "__main__":
    main()
