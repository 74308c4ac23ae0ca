"""Register / LDS / scratch table of every kernel in the fp16 build (the default compute mode), from
`hipcc -Rpass-analysis=kernel-resource-usage` with the product's own flags.  Kernels named in a rocprofv3 kernel_stats.csv (second
argument, optional) are marked as "in the default step".   usage: python tools/resource_table.py [kernel_stats.csv] > table.txt"""
import csv
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd.build import CSRC, FLAGS, HIPCC, SOURCES  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def analyse(src):
    cmd = [HIPCC, *FLAGS, "-DSCOT_OPERAND_FP16", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", os.devnull]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in txt.split("\n"):
        m = re.search(r"remark: .*?(Function Name|VGPRs Spill|SGPRs Spill|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"src": src, "name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(" ")[0] if k not in ("VGPRs Spill", "SGPRs Spill") else k] = int(v)
    return rows


def main():
    used = set()
    if len(sys.argv) > 1:
        for r in csv.DictReader(open(sys.argv[1])):
            used.add(re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", r["Name"])).replace("unsigned short", "bf16").replace("_Float16", "bf16"))
    with ThreadPoolExecutor(8) as ex:
        rows = [r for rs in ex.map(analyse, SOURCES) for r in rs]
    dm = demangle([r["name"] for r in rows])
    seen = set()
    print(f"{'kernel':84s} {'in step':>7s} {'VGPR':>5s} {'AGPR':>5s} {'scratch B/lane':>14s} {'waves/SIMD':>10s} {'LDS B':>7s}")
    for r in sorted(rows, key=lambda r: (r["src"], r["name"])):
        name = re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", dm[r["name"]])).replace("unsigned short", "bf16").replace("_Float16", "bf16")
        if name in seen:
            continue
        seen.add(name)
        mark = "yes" if name in used else ("" if used else "?")
        print(f"{name[:84]:84s} {mark:>7s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('ScratchSize', 0):14d} {r.get('Occupancy', 0):10d} {r.get('LDS', 0):7d}")
    if used:
        bad = [n for n in used if n not in seen]
        print(f"\nkernels of the step not matched by name ({len(bad)}): " + "; ".join(sorted(bad)[:12]))


if __name__ == "__main__":
    main()
