"""Deep-stage GEMM epilogue costs (hot operands: a few MB, as in the step): python tools/bench_deep_gemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd import ops  # noqa: E402


def timeit(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ops.use("f16")
    hd = ops.half_dtype()
    dev = "cuda"
    for M, C in [(1024, 768), (4096, 384)]:
        x = torch.randn(M, C, device=dev).to(hd)
        w1, b1 = (torch.randn(4 * C, C, device=dev) * C ** -0.5).to(hd), torch.randn(4 * C, device=dev)
        w2, b2 = (torch.randn(C, 4 * C, device=dev) * (4 * C) ** -0.5).to(hd), torch.randn(C, device=dev)
        u, gp = torch.empty(M, 4 * C, device=dev, dtype=hd), torch.empty(M, 4 * C, device=dev, dtype=hd)
        y = torch.empty(M, C, device=dev)
        g = torch.randn(M, C, device=dev)
        dy = torch.randn(M, C, device=dev).to(hd)
        du = torch.empty(M, 4 * C, device=dev, dtype=hd)
        w2t, w1t = w2.t().contiguous(), w1.t().contiguous()
        r = {}
        r["fc1 plain"] = timeit(lambda: ops.linear_fwd(ops.BF16, x, w1, u))
        r["fc1 +bias"] = timeit(lambda: ops.linear_fwd(ops.BF16, x, w1, u, bias=b1))
        r["fc1 +bias gelu only"] = timeit(lambda: ops.linear_fwd(ops.BF16, x, w1, u, bias=b1, gelu_deriv_out=u))
        r["fc1 +bias gelu+gelu'"] = timeit(lambda: ops.linear_fwd(ops.BF16, x, w1, u, bias=b1, gelu_deriv_out=gp))
        r["fc2 +bias (fp32 out)"] = timeit(lambda: ops.linear_fwd(ops.BF16, u, w2, y, bias=b2))
        r["dgrad fc2 (*gp)"] = timeit(lambda: ops.linear_dgrad(ops.BF16, dy, w2, du, aux=gp, aux_mul=True, wt=w2t))
        r["dgrad fc2 plain"] = timeit(lambda: ops.linear_dgrad(ops.BF16, dy, w2, du, wt=w2t))
        r["dgrad fc1 (+= g)"] = timeit(lambda: ops.linear_dgrad(ops.BF16, du, w1, g, accumulate=True, wt=w1t))
        print(f"M={M} C={C}: " + " | ".join(f"{k} {v:.1f}" for k, v in r.items()), flush=True)


if __name__ == "__main__":
    main()
