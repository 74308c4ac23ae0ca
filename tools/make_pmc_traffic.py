"""profiles/round3/pmc_traffic.json from the two PMC passes of tools/gpu_r6_evidence.sh (round 3: profiles/round3/gpu_r3_evidence.sh) (tools/pmc_summary.py output files).
usage: python tools/make_pmc_traffic.py <pmc_FETCH_SIZE.txt> <pmc_WRITE_SIZE.txt> [<fetch_total.txt> <write_total.txt>] > pmc_traffic.json"""
import json
import re
import sys

# kernel-name prefix -> the family names bench.py's in-step launch table uses (a launch of the family = one C-ABI call; helper
# kernels of the same call — split-K / plane reduces — are added per call)
FAMILIES = {
    "gemm NT": (r"gemm_fast_kernel<\w+, \d+, \d+, \d+, \d+, \d+, \d+, 0,", ()),
    "gemm NN": (r"gemm_fast_kernel<\w+, \d+, \d+, \d+, \d+, \d+, \d+, 1,", ()),
    "gemm TN": (r"gemm_fast_kernel<\w+, \d+, \d+, \d+, \d+, \d+, \d+, 2,", (r"splitk_",)),
    "wgrad_group": (r"wgrad_group_kernel", (r"wgrad_group_reduce_kernel",)),
    "wgrad_mlp": (r"wgrad_mlp_kernel", (r"plane_reduce_kernel",)),
    "block_tail_fwd": (r"tail_fwd_fused_kernel", ()),
    "block_tail_bwd": (r"tail_bwd_fused_kernel", ()),
    "window_attn_bwd": (r"attn16_bwd_kernel|attn_bwd", ()),
    "window_attn_fwd": (r"attn16_fwd_kernel|attn_fwd", ()),
}


def rows(path):
    out = []
    for line in open(path):
        m = re.match(r"(.*?)\s+grid\s+(\d+)\s+\S+\s+n=\s*(\d+)\s+avg=\s*([\d.]+)", line)
        if m:
            out.append([m.group(1).strip(), int(m.group(2)), int(m.group(3)), float(m.group(4))])
    return out


def per_launch(rs, main, helpers):
    k = [r for r in rs if re.match(main, r[0])]
    n = sum(r[2] for r in k)
    if not n:
        return None
    h = [r for r in rs if any(re.match(p, r[0]) for p in helpers)]
    return (sum(r[2] * r[3] for r in k) + sum(r[2] * r[3] for r in h)) / n


def main():
    f, w = rows(sys.argv[1]), rows(sys.argv[2])
    d = {
        "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace) over `python bench.py --no-graph --steps 3 "
                  "--warmup 1 --no-cpu-baseline --no-parity`; per-dispatch averages by tools/pmc_summary.py (raw: pmc_fetch_size_r<N>.txt, "
                  "pmc_write_size_r<N>.txt beside this file), collected by tools/gpu_r<N>_evidence.sh",
        "units": "counter values are KB; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md (16 B/lane streaming reads are tallied "
                 "at half their bytes); WRITE_SIZE uncalibrated, taken as is",
        "bytes_per_launch": {}, "kb_raw_per_launch": {},
    }
    for fam, (mainp, helpers) in FAMILIES.items():
        fk, wk = per_launch(f, mainp, helpers), per_launch(w, mainp, helpers)
        if fk is None or wk is None:
            continue
        d["kb_raw_per_launch"][fam] = {"fetch": round(fk, 1), "write": round(wk, 1)}
        d["bytes_per_launch"][fam] = int((2 * fk + wk) * 1024)
    d["wgrad_bytes_per_launch"] = d["bytes_per_launch"].get("wgrad_group")
    d["per_kernel_kb_raw"] = {"fetch": f, "write": w}
    if len(sys.argv) > 4:
        tot = {}
        for key, path in (("fetch", sys.argv[3]), ("write", sys.argv[4])):
            m = re.search(r"-> ([\d.]+) per step", open(path).read())
            tot[key + "_kb_raw_per_step"] = float(m.group(1))
        tot["hbm_gb_per_step"] = (2 * tot["fetch_kb_raw_per_step"] + tot["write_kb_raw_per_step"]) * 1024 / 1e9
        tot["note"] = "all dispatches of the run / steps in the run"
        d["whole_step"] = tot
    json.dump(d, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
