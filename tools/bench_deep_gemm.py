"""The Linear layers' NT products at the deep stages of Poseidon-B (batch 64), Poseidon-L (batch 128) or Poseidon-B at 256 x 256 (batch 32):
gemm_fast's 64 x 64 tiles against gemm_wide's 128 x 128 tiles (three instantiations) and the library's policy.  Each configuration is captured as a hipGraph of R back-to-back launches and
replayed, so the number is GPU time per launch incl. the dependent-launch gap (a Python -> ctypes loop cannot issue faster than ~8 us).

    python tools/bench_deep_gemm.py [--model B|L|B256] [--json out.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd import ops  # noqa: E402

R = 20


def graph_time(fn, replays=6):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(R):
            fn()
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
    hd, dev = ops.half_dtype(), "cuda"
    stages = {"B": [(1024, 768), (4096, 384)], "L": [(2048, 1536), (8192, 768), (32768, 384)], "B256": [(2048, 768), (8192, 384)]}[a.model]
    rows = []
    for M, C in stages:
        x = torch.randn(M, C, device=dev).to(hd)
        x4 = torch.randn(M, 4 * C, device=dev).to(hd)
        x3 = torch.randn(M, 3 * C, device=dev).to(hd)
        mk = lambda n, k: (torch.randn(n, k, device=dev) * k ** -0.5).to(hd)
        wqkv, wo, w1, w2 = mk(3 * C, C), mk(C, C), mk(4 * C, C), mk(C, 4 * C)
        w1t, w2t, wqkvt, wot = w1.t().contiguous(), w2.t().contiguous(), wqkv.t().contiguous(), wo.t().contiguous()
        b3, b1, b4 = torch.randn(3 * C, device=dev), torch.randn(C, device=dev), torch.randn(4 * C, device=dev)
        qkv = torch.empty(M, 3 * C, device=dev, dtype=hd)
        u, gp = torch.empty(M, 4 * C, device=dev, dtype=hd), torch.empty(M, 4 * C, device=dev, dtype=hd)
        du = torch.empty(M, 4 * C, device=dev, dtype=hd)
        y32, g32 = torch.empty(M, C, device=dev), torch.randn(M, C, device=dev)
        d16 = torch.empty(M, C, device=dev, dtype=hd)
        cases = [
            ("qkv fwd", 3 * C, C, lambda: ops.linear_fwd(ops.BF16, x, wqkv, qkv, bias=b3)),
            ("proj fwd (fp32 out)", C, C, lambda: ops.linear_fwd(ops.BF16, x, wo, y32, bias=b1)),
            ("fc1 fwd gelu+gelu'", 4 * C, C, lambda: ops.linear_fwd(ops.BF16, x, w1, u, bias=b4, gelu_deriv_out=gp)),
            ("fc2 fwd (fp32 out)", C, 4 * C, lambda: ops.linear_fwd(ops.BF16, x4, w2, y32, bias=b1)),
            ("dgrad fc2 * gelu'", 4 * C, C, lambda: ops.linear_dgrad(ops.BF16, x, w2, du, aux=gp, aux_mul=True, wt=w2t)),
            ("dgrad fc1 += fp32", C, 4 * C, lambda: ops.linear_dgrad(ops.BF16, x4, w1, g32, accumulate=True, wt=w1t)),
            ("dgrad proj", C, C, lambda: ops.linear_dgrad(ops.BF16, x, wo, d16, wt=wot)),
            ("dgrad qkv += fp32", C, 3 * C, lambda: ops.linear_dgrad(ops.BF16, x3, wqkv, g32, accumulate=True, wt=wqkvt)),
        ]
        for name, N, K, fn in cases:
            res = {}
            lib.scot_gemm_wide_config(0, 0)
            res["fast64"] = graph_time(fn)
            lib.scot_gemm_wide_config(1, 0)
            res["policy"] = graph_time(fn)
            for var, vname in ((0, "8w4s"), (1, "8w2s"), (2, "4w2s")):
                lib.scot_gemm_wide_config(2, var)
                res[vname] = graph_time(fn)
            lib.scot_gemm_wide_config(1, 0)
            gf = 2.0 * M * N * K / 1e9
            best = min(res, key=res.get)
            rows.append(dict(M=M, N=N, K=K, name=name, tiles128=(M // 128) * (N // 128), us=res, best=best))
            print(f"M={M:5d} N={N:5d} K={K:5d} {name:22s} tiles128={(M // 128) * (N // 128):4d} | " +
                  " | ".join(f"{k} {v:5.1f}" for k, v in res.items()) + f" | best {best} = {gf / res[best] * 1e3:.0f} TF/s", flush=True)
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
