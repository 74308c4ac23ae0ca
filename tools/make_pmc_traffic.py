"""profiles/round2/pmc_traffic.json from the two PMC passes of tools/gpu_r2d.sh (tools/pmc_summary.py output files).
usage: python tools/make_pmc_traffic.py <pmc_FETCH_SIZE.txt> <pmc_WRITE_SIZE.txt> [<fetch_total.txt> <write_total.txt>] > pmc_traffic.json"""
import json
import re
import sys


def rows(path):
    out = []
    for line in open(path):
        m = re.match(r"(.*?)\s+grid\s+(\d+)\s+\S+\s+n=\s*(\d+)\s+avg=\s*([\d.]+)", line)
        if m:
            out.append([m.group(1).strip(), int(m.group(2)), int(m.group(3)), float(m.group(4))])
    return out


def main():
    f, w = rows(sys.argv[1]), rows(sys.argv[2])

    def per_launch(rs):     # launch-count weighted mean over the wgrad_group_kernel instances (the reduce kernels are added per launch)
        k = [r for r in rs if r[0].startswith("wgrad_group_kernel")]
        red = [r for r in rs if r[0].startswith("wgrad_group_reduce_kernel")]
        n = sum(r[2] for r in k)
        return (sum(r[2] * r[3] for r in k) + sum(r[2] * r[3] for r in red)) / n
    fk, wk = per_launch(f), per_launch(w)
    d = {
        "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace) over `python bench.py --no-graph --steps 3 "
                  "--warmup 1 --no-cpu-baseline --no-parity`; per-dispatch averages by tools/pmc_summary.py (raw: pmc_fetch_size_r2.txt, "
                  "pmc_write_size_r2.txt), collected by tools/gpu_r2d.sh",
        "units": "counter values are KB; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md (16 B/lane streaming reads are tallied "
                 "at half their bytes); WRITE_SIZE uncalibrated, taken as is",
        "wgrad_group_fetch_kb_raw": fk, "wgrad_group_write_kb_raw": wk,
        "wgrad_bytes_per_launch": int((2 * fk + wk) * 1024),
        "wgrad_algorithmic_bytes_per_launch": {"stage0 (K=65536)": 236726592, "stage1 (K=16384)": 129063296,
                                               "note": "16-bit operands read once + fp32 partial tiles written and re-read by the grouped reduce"},
        "per_kernel_kb_raw": {"fetch": f, "write": w},
    }
    if len(sys.argv) > 4:
        tot = {}
        for key, path in (("fetch", sys.argv[3]), ("write", sys.argv[4])):
            m = re.search(r"-> ([\d.]+) per step", open(path).read())
            tot[key + "_kb_raw_per_step"] = float(m.group(1))
        tot["hbm_gb_per_step"] = (2 * tot["fetch_kb_raw_per_step"] + tot["write_kb_raw_per_step"]) * 1024 / 1e9
        tot["note"] = "all dispatches of the run / steps in the run (a step = one cpb_fwd_batched dispatch)"
        d["whole_step"] = tot
    json.dump(d, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
