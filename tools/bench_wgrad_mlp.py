"""Micro-benchmark of scot_wgrad_mlp (cold operands) next to the grouped GEMM it replaces: python tools/bench_wgrad_mlp.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd import ops  # noqa: E402


def main():
    ops.use("f16")
    hd = ops.half_dtype()
    dev = "cuda"
    B = 64
    for L, C in [(1024, 96), (256, 192)]:
        M, hid = B * L, 4 * C
        nset = max(3, int(1.2e9 / (M * C * 4)) + 1)
        w1, b1 = (torch.randn(hid, C, device=dev) * C ** -0.5).to(hd), torch.randn(hid, device=dev) * 0.1
        w2t = (torch.randn(hid, C, device=dev) * hid ** -0.5).to(hd)
        gW = torch.zeros(2 * hid * C + hid + C, device=dev)
        dW1, db1, dW2, db2 = gW[:hid * C].view(hid, C), gW[hid * C:hid * C + hid], gW[hid * C + hid:2 * hid * C + hid].view(C, hid), gW[2 * hid * C + hid:]
        sets = [(torch.randn(M, C, device=dev).to(hd), torch.randn(M, C, device=dev).to(hd)) for _ in range(nset)]
        it = [0]

        def run():
            h, dz = sets[it[0] % nset]
            it[0] += 1
            assert ops.wgrad_mlp(h, dz, w1, b1, w2t, dW1, db1, dW2, db2)
        for _ in range(nset):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3 * nset):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (3 * nset) * 1e3
        print(f"C={C} M={M} wgrad_mlp {us:7.1f} us", end=" | ")
    print()


if __name__ == "__main__":
    main()
