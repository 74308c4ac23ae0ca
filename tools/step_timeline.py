"""Per-queue timeline of ONE training step from a rocprofv3 kernel_trace.csv: where each HIP queue (main chain / side stream) is busy,
which queue finishes last, and what runs at the end of the step.
usage: python tools/step_timeline.py <kernel_trace.csv> [bucket_us=500] [step_index_from_end=2] [marker=patchify]"""
import csv
import sys
from collections import defaultdict, Counter


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    bucket = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 500e3
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qkey], r["Kernel_Name"]) for r in rows)
    marker = sys.argv[4] if len(sys.argv) > 4 else "patchify"      # first kernel of a forward
    ends = [s for s, e, q, n in ks if marker in n.lower()]
    if len(ends) < back + 1:
        print("not enough steps in the trace (marker kernel: %s)" % marker)
        return
    t0, t1 = ends[-back - 1], ends[-back]
    step = [k for k in ks if k[0] >= t0 and k[1] <= t1]
    print(f"step: {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels")
    per = defaultdict(list)
    for k in step:
        per[k[2]].append(k)
    for q, v in sorted(per.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, _, _ in v)
        print(f"queue {q}: {len(v)} kernels, first +{(v[0][0] - t0) / 1e6:.2f} ms, last end +{(max(e for _, e, _, _ in v) - t0) / 1e6:.2f} ms, "
              f"busy {busy / 1e6:.2f} ms")
    nb = int((t1 - t0) / bucket) + 1
    qs = [q for q, _ in sorted(per.items(), key=lambda kv: -len(kv[1]))][:3]
    print("bucket  " + "  ".join(f"q{q:>3} busy% top-kernel{'':22}" for q in qs))
    for b in range(nb):
        lo, hi = t0 + b * bucket, t0 + (b + 1) * bucket
        line = f"{b * bucket / 1e6:6.2f}  "
        for q in qs:
            c, tot = Counter(), 0
            for s, e, _, n in per[q]:
                o = min(e, hi) - max(s, lo)
                if o > 0:
                    c[n.split("<")[0].split("(")[0][-28:]] += o
                    tot += o
            top = c.most_common(1)[0][0] if c else ""
            line += f"{100 * tot / bucket:8.0f}% {top:32}"
        print(line)
    import re
    for q in qs[:2]:
        c, cnt = Counter(), Counter()
        for s_, e, _, n in per[q]:
            key = re.sub(r"\(.*$", "", n.replace("void ", ""))[:60]
            c[key] += e - s_
            cnt[key] += 1
        print(f"queue {q}: kernel time by name (ms per step, launches, us each)")
        for k, v in c.most_common(26):
            print(f"  {v / 1e6:7.3f}  {cnt[k]:4d}  {v / cnt[k] / 1e3:7.1f}  {k}")
    print("last 14 kernels of the step:")
    for s, e, q, n in step[-14:]:
        print(f"  +{(s - t0) / 1e6:7.3f} .. +{(e - t0) / 1e6:7.3f}  q{q}  {n[:70]}")


if __name__ == "__main__":
    main()
