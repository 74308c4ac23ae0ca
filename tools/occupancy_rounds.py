"""Occupancy rounds of every kernel of the default step: workgroups launched / (256 CUs x workgroups a CU holds at once).

A kernel that is bound by its per-workgroup latency chain pays that chain once per ROUND: 1536 workgroups on 512 slots are three rounds,
on 1024 slots 1.5 (the 8x8-window attention backward before / after round 4's register diet: 45 -> 31 us in step).  This table lists, for
each (kernel, grid) of a rocprofv3 trace of `bench.py`, what limits the workgroups per CU (registers, LDS, the 8 waves per SIMD) and how
many rounds the launch is.

  usage: python tools/occupancy_rounds.py profiles/round4/bench_trace_by_grid_r4.txt profiles/round4/pmc_fetch_size_r4.txt > table.txt

Inputs: tools/trace_summary.py's per-grid table (kernel, grid in workgroups, calls per step, average us) and tools/pmc_summary.py's table
(kernel, grid in THREADS) — the two together give the workgroup size; registers / static LDS come from compiling the sources with
-Rpass-analysis=kernel-resource-usage (tools/resource_table.py).  Kernels with DYNAMIC LDS are given below from their launch code."""
import os
import re
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd.build import SOURCES  # noqa: E402
from tools.resource_table import analyse, demangle  # noqa: E402

CUS, LDS_CU, VGPR_SIMD, WAVES_SIMD = 256, 160 * 1024, 512, 8
# dynamic LDS per workgroup (bytes) by kernel-name prefix, 16-bit operands, head_dim 32: csrc/attention_w16.hip / attention.hip launch code
DYNAMIC_LDS = {"attn16_bwd_kernel": 39.6e3, "attn16_fwd_kernel": 2 * 256 * 40 * 2 + 964 * 4, "attn_bwd_kernel<bf16, 32, 4>": 2 * 64 * 40 * 2 + 228 * 12 + 600,
               "attn_fwd_kernel<bf16, 32, 4>": 2 * 64 * 40 * 2 + 228 * 4 + 512, "attn_bwd_kernel<bf16, 32, 2>": 2 * 32 * 40 * 2 + 52 * 12 + 300,
               "attn_fwd_kernel<bf16, 32, 2>": 2 * 32 * 40 * 2 + 52 * 4 + 256}


def norm(name):
    return re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", name)).replace("unsigned short", "bf16").replace("_Float16", "bf16").strip()


def main():
    by_grid, threads = [], {}
    for line in open(sys.argv[1]):
        m = re.match(r"^(.*?)\s+\((\d+), (\d+), (\d+)\)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", line)
        if m:
            by_grid.append((norm(m.group(1)), int(m.group(2)) * int(m.group(3)) * int(m.group(4)), float(m.group(5)), float(m.group(6))))
    for line in open(sys.argv[2]):
        m = re.match(r"^(.*?)\s+grid\s+(\d+)\s", line)
        if m:
            threads.setdefault(norm(m.group(1)), set()).add(int(m.group(2)))
    with ThreadPoolExecutor(8) as ex:
        rows = [r for rs in ex.map(analyse, SOURCES) for r in rs]
    dm = demangle([r["name"] for r in rows])
    res = {norm(dm[r["name"]])[:74]: r for r in rows}
    grids = {}
    for name, wgs, _, _ in by_grid:
        grids.setdefault(name, set()).add(wgs)

    def size_of(name):      # the workgroup size that explains most of this kernel's (workgroups, threads) pairs
        best = max((64, 128, 256, 512, 1024), key=lambda sz: (sum(1 for g in grids[name] if g * sz in threads.get(name, ())), sz == 256))
        return best if any(g * best in threads.get(name, ()) for g in grids[name]) else 256
    print(f"{'kernel':74s} {'wgs':>6s} {'thr':>4s} {'VGPR':>5s} {'LDS KB':>7s} {'wg/CU':>5s} {'limit':>5s} {'rounds':>6s} {'calls':>6s} {'avg us':>7s}")
    for name, wgs, calls, us in by_grid:
        r = res.get(name[:74])
        if r is None or calls < 1:
            continue
        tpw = size_of(name)
        waves = tpw // 64
        regs = r.get("VGPRs", 0) + r.get("AGPRs", 0)
        regs = (regs + 7) // 8 * 8
        per_simd = min(WAVES_SIMD, VGPR_SIMD // max(regs, 8))
        by_regs = per_simd * 4 // waves
        lds = r.get("LDS", 0) or next((v for k, v in DYNAMIC_LDS.items() if name.startswith(k)), 0)
        by_lds = int(LDS_CU // lds) if lds else 99
        per_cu = max(1, min(by_regs, by_lds))
        lim = "LDS" if by_lds < by_regs else ("waves" if per_simd == WAVES_SIMD else "VGPR")
        print(f"{name[:74]:74s} {wgs:6d} {tpw:4d} {regs:5d} {lds / 1024:7.1f} {per_cu:5d} {lim:>5s} {wgs / (CUS * per_cu):6.2f} {calls:6.1f} {us:7.1f}")


if __name__ == "__main__":
    main()
