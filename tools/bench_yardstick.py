"""Yardstick: the library GEMM (torch.matmul = hipBLASLt/rocBLAS) against scot_gemm at every Linear shape of Poseidon-B batch 64,
cold operands, 16-bit operands.  Tells how much headroom a shape has; the product never calls the library GEMM.
  python tools/bench_yardstick.py [f16|bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd import ops  # noqa: E402


def timeit(fn, n):
    for _ in range(n):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3 * n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "f16"
    ops.use(kind)
    hd = ops.half_dtype()
    B = 64
    print(f"{'shape':34s} {'scot us':>9s} {'TF/s':>7s} {'lib us':>9s} {'TF/s':>7s}")
    for s, (L, C) in enumerate([(1024, 96), (256, 192), (64, 384), (16, 768)]):
        M = B * L
        for tag, lay, (m, n, k) in [("qkv fwd", ops.NT, (M, 3 * C, C)), ("fc1 fwd", ops.NT, (M, 4 * C, C)), ("fc2 fwd", ops.NT, (M, C, 4 * C)),
                                    ("proj fwd", ops.NT, (M, C, C)), ("fc2 dgrad", ops.NN, (M, 4 * C, C)), ("fc1 dgrad", ops.NN, (M, C, 4 * C)),
                                    ("qkv dgrad", ops.NN, (M, C, 3 * C)), ("fc1 wgrad", ops.TN, (4 * C, C, M)), ("fc2 wgrad", ops.TN, (C, 4 * C, M)),
                                    ("qkv wgrad", ops.TN, (3 * C, C, M))]:
            per = (m * k + k * n + m * n) * 2
            nset = max(2, min(48, int(1.0e9 / per) + 1))
            sets = []
            for _ in range(nset):
                if lay == ops.NT:
                    a, b = torch.randn(m, k, device="cuda").to(hd), torch.randn(n, k, device="cuda").to(hd)
                elif lay == ops.NN:
                    a, b = torch.randn(m, k, device="cuda").to(hd), torch.randn(k, n, device="cuda").to(hd)
                else:
                    a, b = torch.randn(k, m, device="cuda").to(hd), torch.randn(k, n, device="cuda").to(hd)
                c = torch.zeros(m, n, device="cuda", dtype=torch.float32 if lay == ops.TN else hd)
                sets.append((a, b, c))
            it = [0]

            def mine():
                a, b, c = sets[it[0] % nset]
                it[0] += 1
                ops.gemm(lay, ops.BF16, m, n, k, a, a.shape[1], b, b.shape[1], c, n, accumulate=lay == ops.TN)

            def lib():
                a, b, c = sets[it[0] % nset]
                it[0] += 1
                if lay == ops.NT:
                    torch.matmul(a, b.t(), out=c)
                elif lay == ops.NN:
                    torch.matmul(a, b, out=c)
                else:
                    torch.matmul(a.t(), b)     # 16-bit output (the library has no fp32 += here): traffic-optimistic for the library
            u1, u2 = timeit(mine, nset), timeit(lib, nset)
            fl = 2.0 * m * n * k
            print(f"s{s} {tag:10s} {m:6d}x{n:5d}x{k:6d} {u1:9.1f} {fl / u1 / 1e6:7.1f} {u2:9.1f} {fl / u2 / 1e6:7.1f}")
            del sets


if __name__ == "__main__":
    main()
