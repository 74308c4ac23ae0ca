"""Is a deep-stage K step bound by where its operand slab comes from?  The same 64 x 64-tile NT product (`scot_gemm` through
ops.linear_fwd, fp16 operands) on output shapes from 1 to 192 tiles, K = 768 / 1536 / 3072 / 6144, R back-to-back launches of the SAME operands inside
one hipGraph: with 4 - 16 tiles everything a launch reads (<= 3 MB) stays in every XCD's 4 MB L2 between launches; at 192 tiles (11 MB) it comes from
the Infinity Cache.  Prints us per launch and the slope in ns per 64-deep K step.    python tools/probes/gemm_l2_hot_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from poseidon_amd import ops  # noqa: E402
from tools.bench_deep_gemm import graph_time  # noqa: E402


def main():
    ops.use("f16")
    hd = ops.half_dtype()
    for M, N in [(64, 64), (128, 128), (256, 256), (512, 512), (1024, 768), (4096, 384)]:
        ts = {}
        for K in (768, 1536, 3072, 6144):
            x = torch.randn(M, K, device="cuda").to(hd)
            w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(hd)
            y = torch.empty(M, N, device="cuda", dtype=hd)
            ts[K] = graph_time(lambda: ops.linear_fwd(ops.BF16, x, w, y))
        slope = (ts[6144] - ts[1536]) / ((6144 - 1536) / 64) * 1e3
        mb = (M + N) * 3072 * 2 / 1e6
        print(f"M={M:5d} N={N:4d} tiles={(M // 64) * (N // 64):4d} operands@K3072 {mb:5.1f} MB: " +
              " | ".join(f"K={k}: {v:6.1f} us" for k, v in ts.items()) + f" | {slope:6.0f} ns per K step", flush=True)


if __name__ == "__main__":
    main()
