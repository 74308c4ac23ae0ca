"""Micro-benchmarks of the hot kernels at Poseidon-B batch-64 shapes (HIP-event timed).  python tools/bench_kernels.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd import ops  # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def gemm_case(layout, M, N, K, cdt=torch.bfloat16, out=torch.bfloat16, gelu=False, tag=""):
    cm = ops.BF16 if cdt == torch.bfloat16 else ops.F32
    if layout == ops.NT:
        A, B = torch.randn(M, K, device="cuda").to(cdt), torch.randn(N, K, device="cuda").to(cdt)
    elif layout == ops.NN:
        A, B = torch.randn(M, K, device="cuda").to(cdt), torch.randn(K, N, device="cuda").to(cdt)
    else:
        A, B = torch.randn(K, M, device="cuda").to(cdt), torch.randn(K, N, device="cuda").to(cdt)
        out = torch.float32
    C = torch.zeros(M, N, device="cuda", dtype=out)
    if os.environ.get("BK_COLD", "0") == "1":
        # rotate through enough operand copies to defeat the 256 MB Infinity Cache: in the real step every operand comes
        # from HBM, a loop over ONE buffer set measures the cache-resident rate (2-3x optimistic at stage 0)
        per = A.numel() * A.element_size() + C.numel() * C.element_size() + (B.numel() * B.element_size() if layout == ops.TN else 0)
        nb = max(2, min(64, int(800e6 / per) + 1))
        sets = [(A.clone(), B.clone() if layout == ops.TN else B, C.clone()) for _ in range(nb)]
        it = [0]

        def run():
            a, b, c = sets[it[0] % nb]
            it[0] += 1
            ops.gemm(layout, cm, M, N, K, a, a.shape[1], b, b.shape[1], c, N, a_gelu=gelu, accumulate=layout == ops.TN)
        us = timeit(run, reps=max(20, nb))
    else:
        us = timeit(lambda: ops.gemm(layout, cm, M, N, K, A, A.shape[1], B, B.shape[1], C, N, a_gelu=gelu, accumulate=layout == ops.TN))
    byt = A.numel() * A.element_size() + B.numel() * B.element_size() + C.numel() * C.element_size()
    print(f"gemm {['NT','NN','TN'][layout]} {tag:14s} M={M:6d} N={N:5d} K={K:6d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s  {byt/us/1e3:7.0f} GB/s")


def main():
    ops.L()
    B = 64
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    for s, (L, C) in ([(0, (1024, 96))] if only == "gemm0" else [(1, (256, 192))] if only == "gemm1" else [(2, (64, 384))] if only == "gemm2" else [(3, (16, 768))] if only == "gemm3" else [] if only else list(enumerate([(1024, 96), (256, 192), (64, 384), (16, 768)]))):
        M = B * L
        gemm_case(ops.NT, M, 3 * C, C, tag=f"qkv s{s}")
        gemm_case(ops.NT, M, C, C, out=torch.float32, tag=f"proj s{s}")
        gemm_case(ops.NT, M, 4 * C, C, tag=f"fc1 s{s}")
        gemm_case(ops.NT, M, C, 4 * C, out=torch.float32, gelu=True, tag=f"fc2 s{s}")
        gemm_case(ops.NN, M, C, 4 * C, tag=f"dgrad fc1 s{s}")
        gemm_case(ops.NN, M, 4 * C, C, tag=f"dgrad fc2 s{s}")
        gemm_case(ops.TN, 4 * C, C, M, tag=f"wgrad fc1 s{s}")
        gemm_case(ops.TN, C, 4 * C, M, tag=f"wgrad fc2 s{s}")
    # grouped weight gradients of one ScOTLayer (fc2, fc1, out-projection, qkv), cold operands
    for s_, (L, C) in (list(enumerate([(1024, 96), (256, 192), (64, 384), (16, 768)])) if only == "wgroup" else []):
        K = B * L
        nb = max(2, int(1.2e9 // (K * 12 * C * 2)))     # rotate through > 1 GB of operands
        bf = torch.bfloat16
        sets = []
        for _ in range(nb):
            dy = [torch.randn(K, n, device="cuda").to(bf) for n in (C, 4 * C, C, 3 * C)]
            x = [torch.randn(K, n, device="cuda").to(bf) for n in (4 * C, C, C, C)]
            sets.append((dy, x))
        dw = [torch.zeros(m, n, device="cuda") for m, n in ((C, 4 * C), (4 * C, C), (C, C), (3 * C, C))]
        db = [torch.zeros(m, device="cuda") for m in (C, 4 * C, C, 3 * C)]
        it = [0]

        def run_g():
            dy, x = sets[it[0] % nb]
            it[0] += 1
            assert ops.wgrad_group(ops.BF16, [(dy[i], x[i], dw[i], db[i]) for i in range(4)])
        us = timeit(run_g, reps=30)
        fl = 2.0 * K * 12 * C * C
        print(f"wgrad_group s{s_} K={K} C={C}: {us:7.1f} us  {fl/us/1e6:6.1f} TF/s  {K * 12 * C * 2 / us / 1e6:5.2f} TB/s of operands")
    # CLN
    for L, C in ([(1024, 96), (256, 192), (64, 384), (16, 768)] if only == "cln" else [] if only else [(1024, 96), (256, 192), (16, 768)]):
        rows = B * L
        x, res = torch.randn(rows, C, device="cuda"), torch.randn(rows, C, device="cuda")
        out, out16 = torch.empty_like(x), torch.empty(rows, C, device="cuda", dtype=torch.bfloat16)
        mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
        t = torch.rand(B, device="cuda")
        ps = [torch.randn(C, device="cuda") for _ in range(4)]
        us = timeit(lambda: ops.cln_fwd(x, res, out, mean, rstd, t, ps[0], ps[1], ps[2], ps[3], rows, L, C, 1e-5, out2=out16))
        print(f"cln_fwd rows={rows} C={C}: {us:7.1f} us  {(3*4+2)*rows*C/us/1e3:6.0f} GB/s")
        dx = torch.empty(rows, C, device="cuda", dtype=torch.bfloat16)
        gr = [torch.zeros(C, device="cuda") for _ in range(4)]
        us = timeit(lambda: ops.cln_bwd(out, x, mean, rstd, t, ps[0], ps[1], dx, gr[0], gr[1], gr[2], gr[3], rows, L, C))
        print(f"cln_bwd rows={rows} C={C}: {us:7.1f} us  {(2*4+2)*rows*C/us/1e3:6.0f} GB/s")
    # attention
    for (Hp, C, heads, ws, shift) in ([(32, 96, 3, 16, 8)] if only == "attn" else [(32, 96, 3, 16, 8), (32, 96, 3, 16, 0), (16, 192, 6, 16, 0)] if only == "attn16" else [(4, 768, 24, 4, 0), (8, 384, 12, 8, 0)] if only == "attn3" else [] if only else
                                      [(32, 96, 3, 16, 8), (16, 192, 6, 16, 0), (8, 384, 12, 8, 0), (4, 768, 24, 4, 0)]):
        Lp = Hp * Hp
        qkv = torch.randn(B * Lp, 3 * C, device="cuda").to(torch.bfloat16)
        o = torch.empty(B * Lp, C, device="cuda", dtype=torch.bfloat16)
        nW = (Hp // ws) ** 2
        lse = torch.empty(B * nW, heads, ws * ws, device="cuda")
        tab = torch.randn(heads, (2 * ws - 1) ** 2, device="cuda")
        ls = torch.full((heads,), 2.3, device="cuda")
        us = timeit(lambda: ops.window_attn_fwd(ops.BF16, qkv, o, lse, tab, ls, B, Hp, Hp, C, heads, ws, shift))
        fl = 4.0 * B * nW * heads * (ws * ws) ** 2 * (C // heads)
        print(f"attn_fwd Hp={Hp} C={C} ws={ws}: {us:7.1f} us  {fl/us/1e6:6.1f} TF/s")
        dq = torch.empty_like(qkv)
        dt_, dl = torch.zeros_like(tab), torch.zeros(heads, device="cuda")
        if os.environ.get("BK_COLD", "0") == "1":
            nb = 10
            sets = [(qkv.clone(), o.clone(), torch.randn_like(o), torch.empty_like(qkv)) for _ in range(nb)]
            it = [0]

            def run_f():
                a, b, _, _ = sets[it[0] % nb]
                it[0] += 1
                ops.window_attn_fwd(ops.BF16, a, b, lse, tab, ls, B, Hp, Hp, C, heads, ws, shift)

            def run_b():
                a, b, c, d = sets[it[0] % nb]
                it[0] += 1
                ops.window_attn_bwd(ops.BF16, a, b, c, lse, tab, ls, d, dt_, dl, B, Hp, Hp, C, heads, ws, shift)
            usf, usb = timeit(run_f, reps=30), timeit(run_b, reps=30)
            print(f"attn cold Hp={Hp} C={C} ws={ws}: fwd {usf:7.1f} us  bwd {usb:7.1f} us")
            continue
        us = timeit(lambda: ops.window_attn_bwd(ops.BF16, qkv, o, o, lse, tab, ls, dq, dt_, dl, B, Hp, Hp, C, heads, ws, shift))
        print(f"attn_bwd Hp={Hp} C={C} ws={ws}: {us:7.1f} us  {2.5*fl/us/1e6:6.1f} TF/s")
    # dwconv
    for Hh, C in ([(32, 96)] if only == "dwconv" else [] if only else [(32, 96), (16, 192), (8, 384)]):
        x = torch.randn(B, Hh, Hh, C, device="cuda")
        w, b = torch.randn(C, 49, device="cuda"), torch.randn(C, device="cuda")
        y = torch.empty_like(x)
        us = timeit(lambda: ops.dwconv7(x, w, b, y, B, Hh, Hh, C))
        dw, db = torch.zeros_like(w), torch.zeros_like(b)
        us2 = timeit(lambda: ops.dwconv7_wgrad(y, x, dw, db, B, Hh, Hh, C))
        print(f"dwconv7 H={Hh} C={C}: fwd {us:7.1f} us  wgrad {us2:7.1f} us")


if __name__ == "__main__":
    main()
