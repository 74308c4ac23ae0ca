"""Do the deep stages' GEMMs slow down when their WEIGHTS come from HBM instead of L2 / Infinity Cache?  (In the step every weight matrix is
read once per forward and once per backward — 316 MB of 16-bit copies + as much transposed — so a GEMM's B operand is cold, while
tools/bench_deep_gemm.py re-reads one hot copy.)  Rotates R launches through NW distinct weight copies (NW x bytes >> 256 MB = cold) or one.

    python tools/probe_cold_weights.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd import ops  # noqa: E402
from tools.bench_deep_gemm import graph_time, R  # noqa: E402


def main():
    ops.use("f16")
    hd, dev = ops.half_dtype(), "cuda"
    for M, N, K in [(1024, 3072, 768), (1024, 768, 3072), (1024, 2304, 768), (4096, 1536, 384), (4096, 384, 1536), (4096, 1152, 384)]:
        nw = max(R, int(400e6 // (N * K * 2)) // R * R)
        ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(hd) for _ in range(nw)]
        xs = [torch.randn(M, K, device=dev).to(hd) for _ in range(R)]
        y = torch.empty(M, N, device=dev, dtype=hd)
        b = torch.randn(N, device=dev)
        res = {}
        for label, wsel, xsel in (("hot W, hot X", lambda i: ws[0], lambda i: xs[0]), ("cold W, hot X", lambda i: ws[i % nw], lambda i: xs[0]),
                                  ("cold W, rotating X", lambda i: ws[i % nw], lambda i: xs[i % R])):
            state = {"i": 0}

            def fn():
                i = state["i"]
                state["i"] += 1
                ops.linear_fwd(ops.BF16, xsel(i), wsel(i), y, bias=b)
            # graph_time captures R consecutive calls: the captured launches name R different weight copies; replays re-read them after
            # nw - R other copies have gone through the caches only when nw == R ... so rotate over ALL copies inside one capture instead
            state["i"] = 0
            g = torch.cuda.CUDAGraph()
            fn()
            torch.cuda.synchronize()
            state["i"] = 0
            with torch.cuda.graph(g):
                for _ in range(nw if "cold" in label else R):
                    fn()
            n = nw if "cold" in label else R
            g.replay()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / n * 1e3)
            res[label] = best
        print(f"M={M} N={N} K={K} ({nw} weight copies = {nw * N * K * 2 / 1e6:.0f} MB): " + " | ".join(f"{k} {v:.1f} us" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
