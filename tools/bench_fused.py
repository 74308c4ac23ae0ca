"""Micro-benchmarks of the fused block kernels (csrc/mlp_fused.hip) at Poseidon-B batch-64 shapes, cold operands (buffer sets
rotate through > 1 GB so nothing is served by the 256 MB Infinity Cache), HIP-event timed.
  python tools/bench_fused.py [f16|bf16]
train = stores what the backward consumes (gelu(u), gelu'(u), z, stats); infer = the same kernel without those stores: the
difference is what recomputation in the backward could save in the forward."""
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
    for L, C in [(1024, 96), (256, 192)]:
        M, hid = B * L, 4 * C
        per = M * C * (4 * 4 + 2 * 2) + M * hid * 2 * 2
        nset = max(2, int(1.2e9 / per) + 1)
        dev = "cuda"
        w1, b1 = (torch.randn(hid, C, device=dev) * C ** -0.5).to(hd), torch.randn(hid, device=dev) * 0.1
        w2, b2 = (torch.randn(C, hid, device=dev) * hid ** -0.5).to(hd), torch.randn(C, device=dev) * 0.1
        ps = [torch.randn(C, device=dev) for _ in range(4)]
        t = torch.rand(B, device=dev)
        sets = []
        for _ in range(nset):
            h = torch.randn(M, C, device=dev)
            sets.append(dict(h=h, h16=h.to(hd), out=torch.empty(M, C, device=dev), out16=torch.empty(M, C, device=dev, dtype=hd),
                             act=torch.empty(M, hid, device=dev, dtype=hd), dact=torch.empty(M, hid, device=dev, dtype=hd),
                             z=torch.empty(M, C, device=dev), mean=torch.empty(M, device=dev), rstd=torch.empty(M, device=dev),
                             g=torch.randn(M, C, device=dev), dz=torch.empty(M, C, device=dev, dtype=hd),
                             du=torch.empty(M, hid, device=dev, dtype=hd)))
        it = [0]
        gr = [torch.zeros(C, device=dev) for _ in range(4)]

        def fwd(train):
            s = sets[it[0] % nset]
            it[0] += 1
            ok = ops.mlp_block_fwd(s["h16"], s["h"], w1, b1, w2, b2, s["out"], s["out16"], s["act"] if train else None,
                                   s["dact"] if train else None, s["z"] if train else None, s["mean"] if train else None,
                                   s["rstd"] if train else None, t, ps[0], ps[1], ps[2], ps[3], None, M, L, C, hid, 1e-5)
            assert ok

        def bwd():
            s = sets[it[0] % nset]
            it[0] += 1
            ok = ops.mlp_block_bwd(s["g"], s["g"], s["z"], s["mean"], s["rstd"], t, ps[0], ps[1], None, s["dact"], w1, w2, s["dz"],
                                   s["du"], gr[0], gr[1], gr[2], gr[3], M, L, C, hid)
            assert ok
        fl = 4.0 * M * C * hid
        for name, fn, nb, mult in [("mlp_fwd train", lambda: fwd(True), M * C * (4 + 2 + 4 + 2 + 4) + M * hid * 4, 1),
                                   ("mlp_fwd infer", lambda: fwd(False), M * C * (4 + 2 + 4 + 2), 1),
                                   ("mlp_bwd chain", bwd, M * C * (4 + 4 + 4 + 2) + M * hid * 4, 1)]:
            us = timeit(fn, nset)
            print(f"C={C:3d} M={M:6d} {name:14s}: {us:7.1f} us  {fl * mult / us / 1e6:6.1f} TF/s  {nb / us / 1e3:6.0f} GB/s (algorithmic bytes {nb / 1e6:.0f} MB)")
        # the unfused pieces for reference: fc1 (+gelu, gelu') , fc2, cln
        u, gp, y2 = torch.empty(M, hid, device=dev, dtype=hd), torch.empty(M, hid, device=dev, dtype=hd), torch.empty(M, C, device=dev)
        s = sets[0]
        us1 = timeit(lambda: ops.linear_fwd(ops.BF16, s["h16"], w1, u, bias=b1, gelu_deriv_out=gp), 5)
        us2 = timeit(lambda: ops.linear_fwd(ops.BF16, u, w2, y2, bias=b2), 5)
        print(f"C={C:3d} unfused (hot): fc1+gelu {us1:.1f} us, fc2 {us2:.1f} us")
        del sets


if __name__ == "__main__":
    main()
