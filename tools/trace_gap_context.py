"""The largest idle gaps of the main queue inside ONE step of a rocprofv3 kernel trace, with what ran before / after the gap on the main queue
and what the other queues ran DURING it.   python tools/trace_gap_context.py trace.csv [step_marker_kernel] [n_gaps]"""
import csv
import re
import sys


def short(n):
    return re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", n))[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    marker = sys.argv[2] if len(sys.argv) > 2 else "cpb_fwd_batched"
    ngaps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    marks = [r["s"] for r in rows if marker in r["Kernel_Name"]]
    t0, t1 = marks[-3], marks[-2]          # one step in the middle of the timed steps
    step = [r for r in rows if t0 <= r["s"] < t1]
    byq = {}
    for r in step:
        byq.setdefault(r["Queue_Id"], []).append(r)
    main_q = max(byq, key=lambda q: len(byq[q]))
    mq = byq[main_q]
    gaps = []
    for a, b in zip(mq, mq[1:]):
        if b["s"] - a["e"] > 15000:
            gaps.append((b["s"] - a["e"], a, b))
    gaps.sort(key=lambda g: -g[0])
    print(f"step {(t1 - t0) / 1e6:.3f} ms, main queue {main_q}: {len(mq)} kernels, {sum(g[0] for g in gaps) / 1e6:.3f} ms in {len(gaps)} gaps > 15 us")
    for gap, a, b in sorted(gaps[:ngaps], key=lambda g: g[1]["s"]):
        print(f"+{(a['e'] - t0) / 1e6:7.3f} ms  gap {gap / 1e3:7.1f} us   after {short(a['Kernel_Name'])}  |  before {short(b['Kernel_Name'])}")
        for q, ks in byq.items():
            if q == main_q:
                continue
            during = [k for k in ks if k["e"] > a["e"] and k["s"] < b["s"]]
            if during:
                names = {}
                for k in during:
                    names[short(k["Kernel_Name"])[:40]] = names.get(short(k["Kernel_Name"])[:40], 0) + (min(k["e"], b["s"]) - max(k["s"], a["e"])) / 1e3
                print("            q" + q + ": " + ", ".join(f"{n} {us:.0f} us" for n, us in sorted(names.items(), key=lambda kv: -kv[1])[:4])
                      + f"   (last of them ends {(max(k['e'] for k in during) - b['s']) / 1e3:+.1f} us vs the gap's end)")


if __name__ == "__main__":
    main()
