"""the hand-off-free split-K probe.  The K >= 1152 NT products of Poseidon-B's deep stages at batch 64 whose result is fp32
(fc2 forward: result zeroed by the launcher first; fc1 / qkv data gradients: accumulate into the fp32 residual-stream gradient), unsplit
against S = 2, 3, 4, 6, 8 slices adding into the result with fp32 atomics (csrc/gemm_fast.hip, scot_gemm_splitk_config).  Two timings per
configuration: 'alone' = 20 back-to-back launches of the same product (operands L2 / MALL warm), 'cold' = the same with the operands rotating
through enough copies (> 300 MB) that every launch finds them in HBM, as in the step.  Graph-replayed, GPU time per launch incl. the
dependent-launch gap (and the zeroing memset where there is one).

    python tools/bench_splitk.py [--model B|L|B256] [--json out.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd import ops  # noqa: E402

R = 20


def graph_time(fns, replays=6):
    for fn in fns:
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(R):
            fns[i % len(fns)]()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(replays):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / R * 1e3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="B")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    ops.use("f16")
    lib = ops.L()
    lib.scot_gemm_wide_config(0, 0)
    hd, dev = ops.half_dtype(), "cuda"
    stages = {"B": [(1024, 768), (4096, 384)], "L": [(2048, 1536), (8192, 768)], "B256": [(2048, 768), (8192, 384)]}[a.model]
    rows = []
    for M, C in stages:
        for name, N, K, acc in (("fc2 fwd (zeroed fp32 out)", C, 4 * C, False), ("dgrad fc1 += fp32", C, 4 * C, True),
                                ("dgrad qkv += fp32", C, 3 * C, True), ("proj fwd (zeroed fp32 out)", C, C, False)):
            ncopy = max(2, int(320e6 / (2 * (M * K + N * K) + 4 * M * N)) + 1)
            xs = [torch.randn(M, K, device=dev).to(hd) for _ in range(ncopy)]
            ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(hd) for _ in range(ncopy)]
            wts = [w.t().contiguous() for w in ws]
            ys = [torch.zeros(M, N, device=dev) for _ in range(ncopy)]
            bias = torch.randn(N, device=dev)

            def mk(i):
                if acc:
                    return lambda: ops.linear_dgrad(ops.BF16, xs[i], wts[i], ys[i], accumulate=True, wt=ws[i])
                return lambda: ops.linear_fwd(ops.BF16, xs[i], ws[i], ys[i], bias=bias)
            # correctness of every split against the unsplit kernel (fp32 summation order only)
            lib.scot_gemm_splitk_config(-1, 0)
            ys[0].zero_()
            mk(0)()
            ref = ys[0].clone()
            res = {}
            for S in (1, 2, 3, 4, 6, 8):
                lib.scot_gemm_splitk_config(-1 if S == 1 else S, 1)
                ys[0].zero_()
                mk(0)()
                err = float((ys[0] - ref).abs().max() / ref.abs().max())
                assert err < 1e-5, (name, S, err)
                res[f"S{S}"] = (graph_time([mk(0)]), graph_time([mk(i) for i in range(ncopy)]))
            lib.scot_gemm_splitk_config(0, 1)
            gf = 2.0 * M * N * K / 1e9
            best = min(res, key=lambda k: res[k][1])
            rows.append(dict(M=M, N=N, K=K, name=name, us_alone_cold={k: [round(v[0], 1), round(v[1], 1)] for k, v in res.items()}, best_cold=best))
            print(f"M={M:5d} N={N:5d} K={K:5d} {name:26s} | " + " | ".join(f"{k} {v[0]:5.1f}/{v[1]:5.1f}" for k, v in res.items()) +
                  f" | best cold {best} = {gf / res[best][1] * 1e3:.0f} TF/s", flush=True)
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
