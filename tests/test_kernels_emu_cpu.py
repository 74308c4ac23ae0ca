"""The production kernels of poseidon_amd/csrc executed on the CPU: every .hip source compiled as host C++ against the
hipemu shim (tests/hipemu: one std::thread per work-item, wave64 cross-lane ops / MFMA / transposing LDS read emulated) and
called through the SAME C ABI and the same `poseidon_amd.ops` wrappers as on the GPU, with CPU tensors.  Small shapes only (the arithmetic of every MFMA is
redone per lane).  Test infrastructure: the emulated library is built into a temp directory and
never loaded by the product (`poseidon_amd.lib` only ever opens libscot_hip.so; `ops.ptr` rejects CPU tensors outside this
module's monkeypatch).  What a pass means: index algebra, LDS layout and barrier placement of the kernel sources are right
under the fragment conventions of csrc/common.h — nothing about speed.  The GPU parity tests remain the gate."""
import math
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "hipemu"))
from poseidon_amd import ops  # noqa: E402


@pytest.fixture()
def emu(monkeypatch):
    import emu_session
    lib = emu_session.load_emu()
    emu_session.patch_ops(monkeypatch, lib)
    return lib


def full_only(*cases):   # (was a slow-case gate while work-items were OS threads; as fibers the whole list takes seconds)
    return list(cases)


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def rnd(*shape, dtype=torch.float32, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


TOL = {ops.F32: 2e-5, ops.BF16: 6e-3, ops.X3: 5e-5}
DT = {ops.F32: torch.float32, ops.BF16: torch.bfloat16, ops.X3: torch.float32}


@pytest.mark.parametrize("compute,M,N,K", [(ops.F32, 70, 40, 24), (ops.X3, 130, 72, 40), (ops.BF16, 130, 72, 40), (ops.BF16, 256, 96, 96),
                                           (ops.BF16, 128, 288, 96)] + full_only((ops.BF16, 512, 288, 96), (ops.X3, 128, 96, 96)))
def test_linear_fwd_dgrad_wgrad(emu, compute, M, N, K):
    cdt = DT[compute]
    x, w, b = rnd(M, K, dtype=cdt), rnd(N, K, dtype=cdt, scale=K ** -0.5, seed=1), rnd(N, seed=2)
    y = torch.empty(M, N, dtype=cdt)
    gp = torch.empty(M, N, dtype=cdt)
    ops.linear_fwd(compute, x, w, y, bias=b, gelu_deriv_out=gp)           # gelu(u), gelu'(u)
    u = x.double() @ w.double().t() + b.double()
    assert rel(y, torch.nn.functional.gelu(u)) < TOL[compute]
    assert rel(gp, 0.5 * (1 + torch.erf(u / math.sqrt(2))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)) < TOL[compute]
    dy = rnd(M, N, dtype=cdt, seed=3)
    dx = torch.empty(M, K, dtype=cdt)
    ops.linear_dgrad(compute, dy, w, dx)
    assert rel(dx, dy.double() @ w.double()) < TOL[compute]
    g = rnd(M, K, seed=4)
    gref = g.double() + dy.double() @ w.double()
    ops.linear_dgrad(compute, dy, w, g, accumulate=True)
    assert rel(g, gref) < TOL[compute]
    dw, db = torch.zeros(N, K), torch.zeros(N)
    ops.linear_wgrad(compute, dy, x, dw, dbias=db)
    assert rel(dw, dy.double().t() @ x.double()) < TOL[compute]
    assert rel(db, dy.double().sum(0)) < 1e-4


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 128, 256), (128, 256, 448), (256, 384, 128)])
def test_gemm_wide_tiles(emu, M, N, K, variant):
    """csrc/gemm_wide.hip (mode 2 = every eligible call, kernel variant 0 / 1 / 2: 8 waves x 4 LDS stages, 8 x 2, 4 x 2): every epilogue
    form the engine uses on NT products, against fp64 on the rounded operands, and against gemm_fast's 64 x 64 tiles."""
    lib = emu
    x, w, b = rnd(M, K, dtype=torch.bfloat16), rnd(N, K, dtype=torch.bfloat16, scale=K ** -0.5, seed=1), rnd(N, seed=2)
    u = x.double() @ w.double().t() + b.double()
    lib.scot_gemm_wide_config(0, 0)
    y_fast = torch.empty(M, N)
    ops.linear_fwd(ops.BF16, x, w, y_fast, bias=b)
    lib.scot_gemm_wide_config(2, variant)
    try:
        # forward with bias, 16-bit and fp32 results
        y16, y32 = torch.empty(M, N, dtype=torch.bfloat16), torch.full((M, N), float("nan"))
        ops.linear_fwd(ops.BF16, x, w, y16, bias=b)
        ops.linear_fwd(ops.BF16, x, w, y32, bias=b)
        assert rel(y32, u) < 2e-6 and rel(y16, u) < 6e-3 and rel(y32, y_fast) < 1e-6
        # the fc1 form: gelu(u) and gelu'(u) from one pass
        gv, gd = torch.empty(M, N, dtype=torch.bfloat16), torch.empty(M, N, dtype=torch.bfloat16)
        ops.linear_fwd(ops.BF16, x, w, gv, bias=b, gelu_deriv_out=gd)
        assert rel(gv, torch.nn.functional.gelu(u)) < 6e-3
        assert rel(gd, 0.5 * (1 + torch.erf(u / math.sqrt(2))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)) < 6e-3
        # data gradients (NT on the transposed weight copy): * aux, and accumulated into an fp32 tensor
        if K % 128 == 0:
            dy, wt = rnd(M, N, dtype=torch.bfloat16, seed=3), w.t().contiguous()          # dx[M, K] = dy[M, N] @ w[N, K]
            aux = rnd(M, K, dtype=torch.bfloat16, seed=4)
            dx = torch.empty(M, K, dtype=torch.bfloat16)
            ops.linear_dgrad(ops.BF16, dy, w, dx, aux=aux, aux_mul=True, wt=wt)
            assert rel(dx, (dy.double() @ w.double()) * aux.double()) < 6e-3
            g0 = rnd(M, K, seed=5)
            g = g0.clone()
            ops.linear_dgrad(ops.BF16, dy, w, g, accumulate=True, wt=wt)
            assert rel(g, g0.double() + dy.double() @ w.double()) < 1e-6
    finally:
        lib.scot_gemm_wide_config(1, 0)


@pytest.mark.parametrize("M,N,K,S", [(128, 64, 384, 2), (96, 136, 640, 3), (96, 136, 640, 5), (64, 192, 1024, 2)])
def test_gemm_nt_split_k_atomic(emu, M, N, K, S):
    """csrc/gemm_fast.hip, K slices of the fp32-result NT products adding into the result with fp32 atomics (scot_gemm_splitk_config): the
    forward form (bias from slice 0, result zeroed by the launcher — it starts as NaN here) and the accumulating data-gradient form,
    against fp64 on the rounded operands and against the unsplit kernel; ragged M / N edges and a K that the slices do not divide."""
    lib = emu
    x, w, b = rnd(M, K, dtype=torch.bfloat16), rnd(N, K, dtype=torch.bfloat16, scale=K ** -0.5, seed=1), rnd(N, seed=2)
    u = x.double() @ w.double().t() + b.double()
    lib.scot_gemm_wide_config(0, 0)
    lib.scot_gemm_splitk_config(-1, 0)
    y0 = torch.empty(M, N)
    ops.linear_fwd(ops.BF16, x, w, y0, bias=b)
    g0 = rnd(M, N, seed=5)
    try:
        lib.scot_gemm_splitk_config(S, 1)
        y = torch.full((M, N), float("nan"))
        ops.linear_fwd(ops.BF16, x, w, y, bias=b)
        assert rel(y, u) < 2e-6 and rel(y, y0) < 1e-6
        g = g0.clone()
        ops.linear_dgrad(ops.BF16, x, w.t().contiguous(), g, accumulate=True, wt=w)      # g += x @ (w^T)^T ... the NT product on w itself
        assert rel(g, g0.double() + x.double() @ w.double().t()) < 1e-6
        # accumulating calls only: a non-accumulating call stays unsplit and bit-identical to the unsplit kernel
        lib.scot_gemm_splitk_config(S, 0)
        y = torch.full((M, N), float("nan"))
        ops.linear_fwd(ops.BF16, x, w, y, bias=b)
        assert torch.equal(y, y0)
        # 16-bit results and epilogues beyond the bias are never split
        lib.scot_gemm_splitk_config(S, 1)
        y16a, y16b = torch.empty(M, N, dtype=torch.bfloat16), torch.empty(M, N, dtype=torch.bfloat16)
        ops.linear_fwd(ops.BF16, x, w, y16a, bias=b)
        lib.scot_gemm_splitk_config(-1, 0)
        ops.linear_fwd(ops.BF16, x, w, y16b, bias=b)
        assert torch.equal(y16a, y16b)
    finally:
        lib.scot_gemm_splitk_config(0, 1)
        lib.scot_gemm_wide_config(1, 0)


def test_wgrad_split_k(emu):
    """TN layout with split-K partials + reduce pass (many token rows, small output)."""
    M, N, K = 2048, 96, 96
    dy, x = rnd(M, N, dtype=torch.bfloat16), rnd(M, K, dtype=torch.bfloat16, seed=1)
    dw, db = rnd(N, K, seed=2), torch.zeros(N)
    ref = dw.double() + dy.double().t() @ x.double()
    ops.linear_wgrad(ops.BF16, dy, x, dw, dbias=db)
    assert rel(dw, ref) < 1e-5
    assert rel(db, dy.double().sum(0)) < 1e-4


@pytest.mark.parametrize("K,dims", [(2048, [(96, 384), (384, 96), (96, 96), (288, 96)]),       # a stage-0 block: 96x96 tiles, K split
                                    (8192, [(96, 192), (192, 96)]),                            # t96 policy needs K >= 8192: 4 tiles -> many slices
                                    (256, [(128, 512), (512, 128), (128, 128), (384, 128)]),   # deep stage: 64x64 tiles, no split
                                    (512, [(40, 72), (72, 40), (48, 48)])])                    # ragged tiles (Poseidon-T-like widths)
def test_wgrad_group(emu, K, dims):
    """scot_wgrad_group: the weight (and bias) gradients of several Linear layers over the same K tokens in one launch + one
    grouped reduce == the single-problem launches == fp64."""
    probs, refs = [], []
    for i, (M, N) in enumerate(dims):
        dy, x = rnd(K, M, dtype=torch.bfloat16, seed=10 + i), rnd(K, N, dtype=torch.bfloat16, seed=20 + i)
        dw, db = rnd(M, N, seed=30 + i), (rnd(M, seed=40 + i) if i != 2 else None)
        refs.append((dw.double() + dy.double().t() @ x.double(), None if db is None else db.double() + dy.double().sum(0)))
        probs.append((dy, x, dw, db))
    assert ops.wgrad_group(ops.BF16, probs)
    for (dy, x, dw, db), (rw, rb) in zip(probs, refs):
        assert rel(dw, rw) < 1e-5
        if db is not None:
            assert rel(db, rb) < 1e-4
    # not covered: fp32 operands -> False, nothing written
    assert not ops.wgrad_group(ops.F32, [(p[0].float(), p[1].float(), p[2], p[3]) for p in probs])


@pytest.mark.parametrize("K,dims,forced", [(2048, [(96, 384), (384, 96), (96, 96), (288, 96)], None),      # 96x96 tiles, K slices + grouped reduce
                                           (256, [(128, 512), (512, 128), (128, 128), (384, 128)], None),  # 64x64 tiles, unsplit (single owner)
                                           (256, [(128, 256), (256, 128), (128, 128)], 1),                 # 128x128 tiles, unsplit
                                           (256, [(128, 256), (256, 128), (128, 128)], 1 | (2 << 4)),      # 128x128 tiles, 2 K slices
                                           (512, [(72, 40)], None)])                                       # one ragged problem
def test_wgrad_group_store_and_scaled_modes(emu, K, dims, forced):
    """scot_wgrad_group's `modes` (round 6, lazy zero-grad): 1 = the first writer STORES s·acc over whatever the tensor held (NaN here),
    2 = adds s·acc to unscaled contents, 0 = the plain accumulation; bias gradients always accumulate unscaled.  Every kernel family that
    finishes a grouped weight gradient: single-owner epilogue of the 64x64 / 128x128 tiles, the grouped split-K reduce."""
    lib = emu
    s = torch.tensor([0.125])
    probs, refs, modes = [], [], []
    for i, (M, N) in enumerate(dims):
        dy, x = rnd(K, M, dtype=torch.bfloat16, seed=10 + i), rnd(K, N, dtype=torch.bfloat16, seed=20 + i)
        mode = (ops.GRAD_STORE_SCALED, ops.GRAD_ADD_SCALED, ops.GRAD_ADD)[i % 3]
        dw0 = rnd(M, N, seed=30 + i)
        dw = torch.full((M, N), float("nan")) if mode == ops.GRAD_STORE_SCALED else dw0.clone()
        db = rnd(M, seed=40 + i)
        prod = dy.double().t() @ x.double()
        refs.append(({ops.GRAD_STORE_SCALED: 0.125 * prod, ops.GRAD_ADD_SCALED: dw0.double() + 0.125 * prod, ops.GRAD_ADD: dw0.double() + prod}[mode],
                     db.double() + dy.double().sum(0)))
        probs.append((dy, x, dw, db))
        modes.append(mode)
    if forced is not None:
        lib.scot_gemm_wide_config(2, forced)
    try:
        assert ops.wgrad_group(ops.BF16, probs, modes, s)
    finally:
        lib.scot_gemm_wide_config(1, 0)
    for (dy, x, dw, db), (rw, rb) in zip(probs, refs):
        assert rel(dw, rw) < 1e-5 and rel(db, rb) < 1e-4
    # without a scale tensor the factor is 1
    dy, x, _, _ = probs[0]
    dw = torch.full((dims[0][0], dims[0][1]), float("nan"))
    assert ops.wgrad_group(ops.BF16, [(dy, x, dw, None)], [ops.GRAD_STORE_SCALED], None)
    assert rel(dw, dy.double().t() @ x.double()) < 1e-5


def test_segments_scale_and_fill(emu):
    """scot_segments_scale: a list of (offset, count) pieces of one flat tensor scaled by a device factor (non-finite results counted) or zeroed,
    everything else untouched."""
    x0 = rnd(3 * 4096 + 640)
    segs = [(0, 64), (128, 4096), (4096 + 512, 1000 * 4), (3 * 4096 + 576, 61)]      # (the last piece ends inside a 4-float group: a range's last tensor)
    chunks = torch.tensor([v for sg in segs for v in sg], dtype=torch.int64).reshape(-1, 2)
    x, cnt = x0.clone(), torch.zeros(1, dtype=torch.int32)
    ops.segments_scale(x, chunks, len(segs), torch.tensor([4.0]), cnt)
    ref = x0.clone()
    for o, n in segs:
        ref[o:o + n] *= 4.0
    assert torch.equal(x, ref) and int(cnt) == 0
    x[130] = float("inf")
    ops.segments_scale(x, chunks, len(segs), torch.tensor([0.25]), cnt)
    assert int(cnt) == 1
    ops.segments_scale(x, chunks, len(segs), None)
    for o, n in segs:
        ref[o:o + n] = 0.0
    assert torch.equal(x, ref)


@pytest.mark.parametrize("forced", [0, 1, 2, 1 | (2 << 4), 2 | (3 << 4)])
def test_wgrad_group_wide_tiles(emu, forced):
    """csrc/wgrad_wide.hip: the grouped weight gradients on 128 x 128 tiles — K-strided operands in swizzled [64 tokens][128] LDS tiles read with
    the transposing fragment read, bias gradients from the dY tile, unsplit (read-modify-write by the tile's only writer) and with K slices
    through the grouped reduce.  forced = kernel instantiation | K slices << 4 (scot_gemm_wide_config mode 2)."""
    lib = emu
    K, dims = 256, [(128, 256), (256, 128), (128, 128), (384, 128)]
    probs, refs = [], []
    for i, (M, N) in enumerate(dims):
        dy, x = rnd(K, M, dtype=torch.bfloat16, seed=10 + i), rnd(K, N, dtype=torch.bfloat16, seed=20 + i)
        dw, db = rnd(M, N, seed=30 + i), (rnd(M, seed=40 + i) if i != 2 else None)
        refs.append((dw.double() + dy.double().t() @ x.double(), None if db is None else db.double() + dy.double().sum(0)))
        probs.append((dy, x, dw, db))
    lib.scot_gemm_wide_config(2, forced)
    try:
        assert ops.wgrad_group(ops.BF16, probs)
    finally:
        lib.scot_gemm_wide_config(1, 0)
    for (dy, x, dw, db), (rw, rb) in zip(probs, refs):
        assert rel(dw, rw) < 1e-5
        if db is not None:
            assert rel(db, rb) < 1e-4


@pytest.mark.parametrize("cond", [True, False])
@pytest.mark.parametrize("xdt,B,L,C", [(torch.float32, 2, 40, 96), (torch.bfloat16, 2, 64, 192), (torch.float32, 3, 9, 20), (torch.float32, 3, 16, 768),
                                       (torch.float32, 2, 6, 384)])
def test_cln_fwd_bwd(emu, cond, xdt, B, L, C):
    x, res, t = rnd(B, L, C, dtype=xdt), rnd(B, L, C, seed=1), torch.rand(B)
    gw_w, gw_b, bw_w, bw_b = rnd(C, seed=2, scale=0.3), 1 + rnd(C, seed=3, scale=0.1), rnd(C, seed=4, scale=0.1), rnd(C, seed=5, scale=0.1)
    sc = torch.tensor([1.0 / 0.7, 0.0, 1.0 / 0.7][:B])
    out, out16 = torch.empty(B, L, C), torch.empty(B, L, C, dtype=torch.bfloat16)
    mean, rstd = torch.empty(B * L), torch.empty(B * L)
    ops.cln_fwd(x, res, out, mean, rstd, t if cond else None, gw_w if cond else None, gw_b, bw_w if cond else None, bw_b,
                B * L, L, C, 1e-5, out2=out16, sample_scale=sc)
    dout = rnd(B, L, C, seed=6)
    dx = torch.empty(B, L, C, dtype=xdt)
    grads = [torch.zeros(C) for _ in range(4)]
    ops.cln_bwd(dout, x, mean, rstd, t if cond else None, gw_w if cond else None, gw_b, dx, grads[0], grads[1], grads[2], grads[3],
                B * L, L, C, sample_scale=sc)
    x64 = x.double().requires_grad_(True)
    ps = [p.double().requires_grad_(True) for p in (gw_w, gw_b, bw_w, bw_b)]
    mu = x64.mean(-1, keepdim=True)
    xh = (x64 - mu) / torch.sqrt(x64.var(-1, unbiased=False, keepdim=True) + 1e-5)
    g = t.double().view(B, 1, 1) * ps[0] + ps[1] if cond else ps[1]
    b = t.double().view(B, 1, 1) * ps[2] + ps[3] if cond else ps[3]
    ref = res.double() + sc.double().view(B, 1, 1) * (g * xh + b)
    ref.backward(dout.double())
    assert rel(out, ref.detach()) < 1e-5 and torch.equal(out16, out.to(torch.bfloat16))
    assert rel(dx, x64.grad) < (1e-4 if xdt == torch.float32 else 1e-2)
    for i in ([0, 1, 2, 3] if cond else [1, 3]):
        assert rel(grads[i], ps[i].grad) < 1e-4, i

    # mode 3 (the deep stages' form): dx + per-block partial sums, finished into CONTIGUOUS parameter gradients by a second launch
    nf = ops.cln_bwd_partial_floats(B * L, L, C, cond)
    assert (nf > 0) == (C % 64 == 0 and 128 <= C <= 1536 and B * L <= 8192)
    if nf:
        part = torch.full((nf,), float("nan"))
        dx3 = torch.empty(B, L, C, dtype=xdt)
        flat = torch.zeros(4 * C) + 0.25                      # += semantics: starts non-zero
        g3 = [flat[i * C:(i + 1) * C] for i in range(4)] if cond else [None, flat[0:C], None, flat[C:2 * C]]
        ops.cln_bwd(dout, x, mean, rstd, t if cond else None, gw_w if cond else None, gw_b, dx3, None, None, None, None, B * L, L, C,
                    sample_scale=sc, mode=3, partial=part)
        ops.cln_bwd_finish(part, B * L, L, C, g3[0], g3[1], g3[2], g3[3])
        pass
        assert torch.equal(dx3, dx) or rel(dx3, dx) < 1e-6
        for i in ([0, 1, 2, 3] if cond else [1, 3]):
            assert rel(g3[i] - 0.25, ps[i].grad) < 1e-4, i


ATTN = [  # compute, B, Hp, Wp, C, heads, ws, shift
    (ops.BF16, 1, 16, 16, 32, 1, 16, 0),     # 16x16-window fast path, one window, head_dim 32
    (ops.BF16, 1, 32, 32, 32, 1, 16, 8),     # ... shifted: all mask regions (4 windows)
    (ops.BF16, 1, 16, 16, 32, 2, 16, 0),     # ... head_dim 16 (one-pass backward with a single feature block)
    (ops.BF16, 2, 8, 8, 32, 2, 4, 2),        # general kernels, N = 16, shifted
    (ops.F32, 1, 7, 7, 16, 1, 7, 0),         # general kernels, N = 49 (ragged tiles), exact fp32 MFMA
] + full_only((ops.X3, 1, 16, 16, 16, 1, 16, 0))   # 16x16 fast path with hi/lo split operands, head_dim 16


@pytest.mark.parametrize("case", ATTN, ids=lambda c: "-".join(str(x) for x in c))
def test_window_attention_fwd_bwd(emu, case):
    from test_kernels_gpu import _attn_ref
    compute, B, Hp, Wp, C, heads, ws, shift = case
    cdt = DT[compute]
    L, TS, N = Hp * Wp, (2 * ws - 1) ** 2, ws * ws
    qkv = rnd(B, L, 3 * C, dtype=cdt)
    table = (16 * torch.sigmoid(rnd(heads, TS, seed=1))).contiguous()
    ls = torch.linspace(math.log(3.0), math.log(20.0), heads)
    dout = rnd(B, L, C, dtype=cdt, seed=2)
    out = torch.full((B, L, C), float("nan"), dtype=cdt)
    nW = (Hp // ws) * (Wp // ws)
    lse = torch.empty(B * nW, heads, N)
    ops.window_attn_fwd(compute, qkv, out, lse, table, ls, B, Hp, Wp, C, heads, ws, shift)
    dqkv = torch.full((B, L, 3 * C), float("nan"), dtype=cdt)
    dtab, dls = torch.zeros(heads, TS), torch.zeros(heads)
    ops.window_attn_bwd(compute, qkv, out, dout, lse, table, ls, dqkv, dtab, dls, B, Hp, Wp, C, heads, ws, shift)
    q64, t64, l64 = qkv.double().requires_grad_(True), table.double().requires_grad_(True), ls.double().requires_grad_(True)
    ref = _attn_ref(q64, t64, l64, B, Hp, Wp, C, heads, ws, shift)
    ref.backward(dout.double())
    tol_o, tol_g = (2e-5, 5e-5) if compute == ops.F32 else (5e-5, 2e-4) if compute == ops.X3 else (2e-2, 4e-2)
    assert rel(out, ref.detach()) < tol_o
    assert rel(dqkv, q64.grad) < tol_g
    assert rel(dtab, t64.grad) < tol_g
    assert rel(dls, l64.grad) < (2e-4 if compute == ops.F32 else 2e-3 if compute == ops.X3 else 0.15)
    # replica entry (window w -> replica w % R) + the fold of the replicas: the same sums
    R, st, sl = 2, heads * TS + 3, heads + 1
    rt, rl, dq2, dst = torch.zeros(R * st), torch.zeros(R * sl), torch.empty_like(dqkv), torch.zeros(heads + 2)
    ops.window_attn_bwd_rep(compute, qkv, out, dout, lse, table, ls, dq2, rt, rl, B, Hp, Wp, C, heads, ws, shift, R, st, sl)
    assert torch.equal(dq2, dqkv) and (B * nW > 1) == bool(rt[st:].abs().sum() > 0)
    ops.replica_reduce(rt, 1, R, st, torch.tensor([0, 0, heads * TS], dtype=torch.int32), 1, heads * TS, rt)
    ops.replica_reduce(rl, 0, R, sl, torch.tensor([0, 2, heads], dtype=torch.int32), 1, heads, dst)
    assert rel(rt[:heads * TS].view(heads, TS), dtab) < 1e-5 and rel(dst[2:], dls) < 5e-4 and torch.all(dst[:2] == 0)


# ---- the fused block kernels of csrc/mlp_fused.hip: the bodies of their (still gated) GPU parity tests, run here on CPU tensors
FUSED = [(1, 72, 96), (1, 64, 192)] + full_only((2, 128, 96), (3, 72, 96))


@pytest.fixture()
def gpu_test_bodies(emu, monkeypatch):
    import test_kernels_gpu as G
    monkeypatch.setattr(G, "DEV", "cpu")
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    return G


@pytest.mark.parametrize("B,L,C", FUSED)
@pytest.mark.parametrize("train,cond", [(True, True), (False, False)])
def test_fused_mlp_and_projection_forward(gpu_test_bodies, train, cond, B, L, C):
    gpu_test_bodies.test_mlp_block_fused(train, cond, B, L, C)
    gpu_test_bodies.test_proj_cln_fused(train, cond, B, L, C)
    gpu_test_bodies.test_block_tail_fwd_fused(train, cond, B, L, C, next_qkv=True)


@pytest.mark.parametrize("train,cond,B,L", [(True, True, 1, 72), (False, False, 2, 64)])
def test_fused_tail_forward_c48(gpu_test_bodies, train, cond, B, L):
    gpu_test_bodies.test_block_tail_fwd_fused_c48(train, cond, B, L, next_qkv=True)


@pytest.mark.parametrize("cond,B,L", [(True, 1, 64), (False, 2, 64)])
def test_fused_tail_backward_c48(gpu_test_bodies, cond, B, L):
    gpu_test_bodies.test_block_tail_bwd_fused_c48(cond, B, L)


@pytest.mark.parametrize("cond,B,L,C", [(True, 1, 64, 96), (False, 1, 64, 192)] + full_only((False, 2, 128, 96), (True, 3, 64, 96),
                                                                                          (True, 1, 64, 192)))
def test_fused_mlp_and_projection_backward(gpu_test_bodies, cond, B, L, C):
    gpu_test_bodies.test_mlp_block_bwd_fused(cond, B, L, C)
    gpu_test_bodies.test_proj_cln_bwd_fused(cond, B, L, C)
    gpu_test_bodies.test_block_tail_bwd_fused(cond, B, L, C, False)
    gpu_test_bodies.test_block_tail_bwd_fused(cond, B, L, C, True)


@pytest.mark.parametrize("cond,B,L,C,prologue", [(True, 1, 64, 96, True), (False, 1, 64, 192, False), (True, 3, 64, 96, False), (True, 2, 64, 192, True)])
def test_block_tail_lean_forms(gpu_test_bodies, cond, B, L, C, prologue):
    gpu_test_bodies.test_block_tail_lean_forms(cond, B, L, C, prologue)


@pytest.mark.parametrize("s,t", [(32, 64), (32, 16), (64, 32), (24, 40)])
def test_spectral_resize_native(emu, s, t):
    """scOT.model.spectral_resize (NT GEMM on the fp32 MFMA + scot_spectral_apply) == the reference's fft2 -> crop / pad -> ifft2
    (model.py:1293-1316, restated in oracle.scot_cpu.spectral_resize with torch.fft), forward and gradient."""
    from oracle.scot_cpu import spectral_resize as ref_resize
    from scOT.model import _SpectralResize
    x = rnd(2, 3, s, s).requires_grad_(True)
    y = _SpectralResize.apply(x, t)
    xr = x.detach().double().requires_grad_(True)
    yr = ref_resize(xr, t)
    assert tuple(y.shape) == (2, 3, t, t) and rel(y.detach(), yr.detach()) < 2e-6
    g = rnd(2, 3, t, t, seed=5)
    y.backward(g)
    yr.backward(g.double())
    assert rel(x.grad, xr.grad) < 2e-6


def test_cpb_batched_layers(gpu_test_bodies):
    gpu_test_bodies.test_cpb_batched_layers()


def test_transposed_weight_copies(gpu_test_bodies):
    gpu_test_bodies.test_transpose_cast_and_dgrad_nt()


@pytest.mark.parametrize("H,W,C", [(8, 8, 6), (9, 5, 16), (8, 8, 96)])
def test_space_to_depth_shuffles(gpu_test_bodies, H, W, C):
    gpu_test_bodies.test_space_depth(H, W, C)


@pytest.mark.parametrize("H,W,C,B", [(16, 16, 96, 2), (5, 5, 24, 2)])
def test_depthwise_conv7(gpu_test_bodies, H, W, C, B):
    gpu_test_bodies.test_dwconv7(H, W, C, B)


@pytest.mark.parametrize("H,W,Cc", [(16, 16, 3), (18, 14, 3), (32, 64, 4)])
def test_patchify_strip_kernel(gpu_test_bodies, H, W, Cc):
    gpu_test_bodies.test_patchify_unpatchify(H, W, Cc, torch.float32)


@pytest.mark.parametrize("Cc,H,W", [(4, 32, 32), (5, 13, 10), (1, 20, 132)])
def test_conv5_tiled(gpu_test_bodies, Cc, H, W):
    gpu_test_bodies.test_conv5(Cc, H, W)


@pytest.mark.parametrize("M,N,dt", [(4099, 48, torch.float32), (1024, 768, torch.bfloat16), (300, 2048, torch.float32), (129, 8, torch.bfloat16),
                                    (777, 200, torch.float32), (513, 100, torch.bfloat16)])
def test_colsum_vector_and_scalar_forms(gpu_test_bodies, M, N, dt):
    """scot_colsum: the 8-column vector kernel (N % 8 == 0; round 6) and the scalar one, ragged row counts"""
    gpu_test_bodies.test_colsum_shapes(M, N, dt)
    gpu_test_bodies.test_reductions_and_scale_residual()


def test_optimizer_kernels_skip_clock_scale_and_operand_copy(emu):
    """csrc/optim.hip on the CPU: scot_clip_coef's non-finite flag, scot_adamw_step (torch.optim.AdamW arithmetic, Adam's clock read
    from the device, skip on a non-finite norm, 16-bit operand copy of the new weights in the same pass), scot_optim_finish
    (GradScaler's scale schedule) and scot_scale_inplace_dev (the scale read from the device)."""
    import ctypes
    import numpy as np
    L = ops.L()
    n = 64 * 24
    g0 = torch.Generator().manual_seed(0)
    p = torch.randn(n, generator=g0)
    ref = p.clone().requires_grad_(True)
    topt = torch.optim.AdamW([ref], lr=1e-2, weight_decay=0.1, betas=(0.9, 0.999), eps=1e-8)
    m, v = torch.zeros(n), torch.zeros(n)
    map8 = torch.zeros(n // 8, dtype=torch.uint8)
    map8[-2:] = 255                                               # alignment padding: never stepped, never copied
    nblk = int(L.scot_optim_blocks(n))
    partial, clip = torch.empty(nblk), torch.tensor([1.0, 0.0, 0.0])
    step_state = torch.zeros(2, dtype=torch.int32)
    scale_state = torch.tensor([1024.0, 1.0 / 1024.0, 0.0, 0.0])
    shadow = torch.full((n,), -7.0, dtype=torch.bfloat16)
    lr, wd = (ctypes.c_float * 1)(1e-2), (ctypes.c_float * 1)(0.1)

    def step(grad, host_step):
        assert L.scot_grad_sqnorm(grad.data_ptr(), map8.data_ptr(), n, partial.data_ptr(), None) == 0
        assert L.scot_clip_coef(partial.data_ptr(), nblk, 0.0, clip.data_ptr(), None) == 0
        assert L.scot_adamw_step(p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), map8.data_ptr(), n, ctypes.cast(lr, ctypes.c_void_p),
                                 ctypes.cast(wd, ctypes.c_void_p), 1, 0.9, 0.999, 1e-8, host_step, clip.data_ptr(), step_state.data_ptr(),
                                 shadow.data_ptr(), None) == 0
        assert L.scot_optim_finish(step_state.data_ptr(), clip.data_ptr(), scale_state.data_ptr(), 2.0, 0.5, 2, float(2 ** 20), None) == 0

    grads = [torch.randn(n, generator=g0) for _ in range(4)]
    live = slice(0, n - 16)
    for k, g in enumerate(grads[:2]):
        step(g, k + 1)
        ref.grad = g.clone()
        topt.step()
    assert clip.tolist()[0] == 1.0 and clip.tolist()[2] == 0.0 and abs(clip[1].item() - grads[1][live].norm().item()) < 1e-3
    assert torch.allclose(p[live], ref.detach()[live], rtol=1e-6, atol=1e-7) and step_state.tolist() == [2, 0]
    assert torch.equal(shadow[live], p[live].to(torch.bfloat16)) and torch.all(shadow[n - 16:] == -7.0)      # padding untouched
    assert scale_state.tolist()[:3] == [2048.0, 1.0 / 2048.0, 0.0]                                           # two clean steps: doubled
    bad = grads[2].clone()
    bad[5] = float("nan")
    p_before, m_before, sh_before = p.clone(), m.clone(), shadow.clone()
    step(bad, 99)                                                  # the host's step number is ignored: the device clock rules
    assert clip[2].item() == 1.0 and torch.equal(p, p_before) and torch.equal(m, m_before) and torch.equal(shadow, sh_before)
    assert step_state.tolist() == [2, 1] and scale_state.tolist()[:3] == [1024.0, 1.0 / 1024.0, 0.0]
    step(grads[3], 99)                                             # resumes as Adam step 3 (bias correction of the third APPLIED step)
    ref.grad = grads[3].clone()
    topt.step()
    assert torch.allclose(p[live], ref.detach()[live], rtol=1e-6, atol=1e-7) and step_state.tolist() == [3, 1]
    # the device-resident scale: x *= *scale, non-finite results counted
    x = torch.arange(1, 41, dtype=torch.float32)
    cnt = torch.zeros(1, dtype=torch.int32)
    ops.scale_inplace_dev(x, scale_state[1:2], cnt)
    assert torch.equal(x, torch.arange(1, 41, dtype=torch.float32) / 1024.0) and int(cnt) == 0
    x[3] = float("inf")
    ops.scale_inplace_dev(x, scale_state[0:1], cnt)
    assert int(cnt) == 1 and np.isinf(x[3].item())


def test_step_tape_program_is_the_recorded_call_sequence(emu):
    """ops.compile_tape / scot_tape_replay (csrc/host_tape.hip): the program replayed inside the library is, entry by entry, the list of
    C-ABI calls the recorder logged (same entry points, same integer-class and float arguments in order), host-side steps cut it into
    runs, and replaying it reproduces the direct calls — including a 21-argument entry point with a float in the middle (scot_cln_fwd)
    and entry points whose arguments spill to the stack."""
    import ctypes
    import struct
    M, N, K, C = 64, 32, 32, 32
    x, w, b = rnd(M, K, dtype=torch.bfloat16), rnd(N, K, dtype=torch.bfloat16, seed=1), rnd(N, seed=2)
    resid = rnd(M, N, seed=3)
    gw, gb = 1 + rnd(C, seed=4, scale=0.1), rnd(C, seed=5, scale=0.1)

    def run(y, out, out16, mean, rstd, scratch, marks):
        ops.memset_async(scratch, 0x3f)
        ops.linear_fwd(ops.BF16, x, w, y, bias=b)
        marks.append("host step")
        ops.cln_fwd(y, resid, out, mean, rstd, None, None, gw, None, gb, M, M, C, 1e-5, out2=out16)
        ops.memcpy_async(scratch, mean)

    def bufs():
        return (torch.empty(M, N), torch.empty(M, C), torch.empty(M, C, dtype=torch.bfloat16), torch.empty(M), torch.empty(M),
                torch.zeros(M))
    ref = bufs()
    run(*ref, [])
    got = bufs()
    log, marks = [], []
    prev = ops.set_recorder(log)
    try:
        ops.memset_async(got[5], 0x3f)
        ops.linear_fwd(ops.BF16, x, w, got[0], bias=b)
        log.append((lambda: marks.append("host step"), None))
        ops.cln_fwd(got[0], resid, got[1], got[3], got[4], None, None, gw, None, gb, M, M, C, 1e-5, out2=got[2])
        ops.memcpy_async(got[5], got[3])
    finally:
        ops.set_recorder(prev)
    segs = ops.compile_tape(log)
    assert [s[0] for s in segs] == ["c", "py", "c"]
    # decode the words back into (address, ints, floats) and compare with the log
    calls = [(fn, args) for fn, args in log if args is not None]
    decoded = []
    for s in segs:
        if s[0] != "c":
            continue
        words, i = list(s[1]), 0
        while i < s[2]:
            ni, nf = words[i + 1], words[i + 2]
            decoded.append((words[i], words[i + 3:i + 3 + ni], words[i + 3 + ni:i + 3 + ni + nf]))
            i += 3 + ni + nf
    assert len(decoded) == len(calls) == 4
    for (addr, ints, flts), (fn, args) in zip(decoded, calls):
        assert addr == ctypes.cast(fn, ctypes.c_void_p).value
        want_i = [(0 if a is None else a & 0xFFFFFFFFFFFFFFFF) for a, t in zip(args, fn.argtypes) if t is not ctypes.c_float]
        want_f = [struct.unpack("<I", struct.pack("<f", a))[0] for a, t in zip(args, fn.argtypes) if t is ctypes.c_float]
        assert ints == want_i and flts == want_f
    for t in got:
        t.fill_(float("nan")) if t.dtype != torch.bfloat16 else t.zero_()
    marks.clear()
    ops.replay_tape(segs)
    assert marks == ["host step"]
    for a, r in zip(got, ref):
        assert torch.equal(a, r)
