"""Fixed vs per-K-tile cost of the tiled GEMM at the deep stages' output shapes (hot operands): python tools/bench_gemm_k.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd import ops  # noqa: E402
from tools.bench_deep_gemm import timeit  # noqa: E402


def main():
    ops.use("f16")
    hd = ops.half_dtype()
    for M, N in [(4096, 1536), (4096, 384), (1024, 3072), (1024, 768)]:
        row = []
        for K in (64, 128, 256, 384, 768, 1536, 3072):
            x = torch.randn(M, K, device="cuda").to(hd)
            w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(hd)
            y = torch.empty(M, N, device="cuda", dtype=hd)
            row.append(f"K={K}: {timeit(lambda: ops.linear_fwd(ops.BF16, x, w, y)):5.1f}")
        print(f"M={M} N={N} (16-bit out): " + " | ".join(row), flush=True)
    e = torch.empty(64, device="cuda")
    print(f"empty-ish launch (scale_inplace of 64 floats): {timeit(lambda: ops.scale_inplace(e, 1.0)):5.1f} us")


if __name__ == "__main__":
    main()
