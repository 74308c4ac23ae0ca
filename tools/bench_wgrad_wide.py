"""The grouped weight gradients of a deep-stage ScOTLayer (fc2, fc1, projection, qkv over the same K tokens): gemm_fast's 64 x 64-tile grouped
kernel against wgrad_wide's 128 x 128 tiles (three instantiations x K slices), hipGraph replays of 20 launches (tools/bench_deep_gemm.py).

    python tools/bench_wgrad_wide.py [--model B|L|B256]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd import ops  # noqa: E402
from tools.bench_deep_gemm import graph_time  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="B")
    a = ap.parse_args()
    ops.use("f16")
    lib = ops.L()
    hd, dev = ops.half_dtype(), "cuda"
    stages = {"B": [(1024, 768), (4096, 384)], "L": [(2048, 1536), (8192, 768), (32768, 384)], "B256": [(2048, 768), (8192, 384)]}[a.model]
    for K, C in stages:
        dims = [(C, 4 * C), (4 * C, C), (C, C), (3 * C, C)]
        probs = [((torch.randn(K, m, device=dev) * 0.5).to(hd), torch.randn(K, n, device=dev).to(hd), torch.zeros(m, n, device=dev), torch.zeros(m, device=dev))
                 for m, n in dims]
        fn = lambda: ops.wgrad_group(ops.BF16, probs)
        gf = sum(2.0 * K * m * n for m, n in dims) / 1e9
        t128 = sum((m // 128) * (n // 128) for m, n in dims)
        res = {}
        lib.scot_gemm_wide_config(0, 0)
        res["64x64 grouped"] = graph_time(fn)
        lib.scot_gemm_wide_config(1, 0)
        res["policy"] = graph_time(fn)
        for var, vname in ((0, "8w4s"), (1, "8w2s"), (2, "4w2s")):
            for S in (1, 2, 3, 4, 6, 8):
                if S > 1 and (t128 * S > 1200 or K // 64 // S < 4):
                    continue
                lib.scot_gemm_wide_config(2, var | (S << 4))
                res[f"{vname} S={S}"] = graph_time(fn)
        lib.scot_gemm_wide_config(1, 0)
        best = min(res, key=res.get)
        print(f"K={K:6d} C={C:5d} tiles128={t128:5d} | " + " | ".join(f"{k} {v:6.1f}" for k, v in res.items()) +
              f" | best {best} = {gf / res[best] * 1e3:.0f} TF/s (64x64: {gf / res['64x64 grouped'] * 1e3:.0f})", flush=True)


if __name__ == "__main__":
    main()
