"""Thin tensor-level wrappers over the C ABI: pointers, sizes and the current HIP stream go down, nothing else.

Every function enqueues exactly the kernels named in its docstring on torch's current stream and returns
immediately (no synchronisation).  Tensors must live on the GPU and be contiguous; dtype is float32 or bfloat16.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import lib as _lib

F32, BF16, X3 = 0, 1, 2   # compute modes of scot_gemm (X3 = bf16x3: fp32 operands, hi/lo bf16 split, 3 MFMAs per K-step)
NT, NN, TN = 0, 1, 2


_active = "bf16"   # which build of the library the calls below go to: "bf16" (libscot_hip.so) or "f16" (libscot_hip_f16.so)
HALF = {"bf16": torch.bfloat16, "f16": torch.float16}


def use(kind: str) -> str:
    """Route the wrappers below to the library build whose 16-bit operand format is `kind`; returns the previous one.
    (An engine selects its build around every forward / backward; a recorded step tape holds the functions themselves.)"""
    global _active
    if kind not in HALF:
        raise ValueError(kind)
    prev, _active = _active, kind
    return prev


def half_dtype() -> torch.dtype:
    return HALF[_active]


def dt(t: torch.Tensor) -> int:
    """dtype code of the C ABI: 0 = float32, 1 = the active library's 16-bit operand format."""
    if t.dtype == torch.float32:
        return F32
    if t.dtype == HALF[_active]:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype} for the {_active} build of the library")


def ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.ScotLibraryError("scOT HIP ops need GPU tensors (no CPU path exists in the product)")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


_selftested = set()


_recorder = None   # a list while the engine records a step tape: every C-ABI call is appended as (function, args)


class _Recording:
    """Library proxy used while a step is being recorded: calls run as usual AND are logged with their final arguments
    (device addresses, sizes, stream handles) so that `replay` can re-issue exactly the same launches."""

    def __init__(self, lib, log):
        self._lib, self._log = lib, log

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        log = self._log

        def call(*args):
            rc = fn(*args)
            if rc != -3:      # SCOT_ERR_UNSUPPORTED = "not covered, nothing launched": the caller falls back, the replay must not re-ask
                log.append((fn, args))
            return rc
        return call


def set_recorder(log):
    """log: list to append (fn, args) to, or None to stop recording.  Returns the previous recorder."""
    global _recorder
    prev, _recorder = _recorder, log
    return prev


def L():
    """Library handle; runs the one-time device self test of the transposing LDS read on first use."""
    l = _lib.load(kind=_active)
    if _active not in _selftested and torch.cuda.is_available():
        rc = l.scot_selftest_tr(stream())
        if rc < 0:
            raise _lib.ScotLibraryError("scot_selftest_tr failed to run")
        _selftested.add(_active)
    if _recorder is not None:
        return _Recording(l, _recorder)
    return l


_workspace = {}
_retired = []                   # outgrown scratch buffers: kept alive (launches already enqueued may still use them)
WORKSPACE_MIN_BYTES = 8 << 20


_slot = 0


def set_workspace_slot(slot: int) -> int:
    """0 = main stream, 1 = side stream, 2.. = batch-chain streams: kernels of different streams run concurrently and must
    not share split-K scratch.  Returns the previous slot."""
    global _slot
    prev, _slot = _slot, slot
    return prev


def workspace(need: int = 0):
    """Per-(device, slot) scratch for split-K partial tiles, sized by the library's own `scot_*_workspace_bytes` answers for the
    shapes actually launched (grown to the next power of two on first need; the C ABI never allocates).  A recorded step tape holds
    the address it was recorded with: growth only happens on a shape's first (eager, unrecorded) call."""
    key = (torch.cuda.current_device(), _slot)
    w = _workspace.get(key)
    if w is None or w.numel() < need:
        size = WORKSPACE_MIN_BYTES
        while size < need:
            size *= 2
        if w is not None:
            _retired.append(w)
        w = torch.empty(size, dtype=torch.uint8, device=f"cuda:{key[0]}")
        _workspace[key] = w
    return w


def _raw():
    """the library itself, never the recording proxy (size queries are not launches)"""
    l = L()
    return l._lib if isinstance(l, _Recording) else l


def _gemm_ws(layout, compute, M, N, K):
    return workspace(int(_raw().scot_gemm_workspace_bytes(layout, compute, M, N, K)))


def gemm(layout: int, compute: int, M: int, N: int, K: int, A, lda: int, B, ldb: int, C, ldc: int, *, bias=None,
         colscale=None, aux=None, ldaux: int = 0, resid=None, ldres: int = 0, a_gelu: bool = False, b_gelu: bool = False,
         accumulate: bool = False, colsum_out=None, aux_mul: bool = False, gelu_deriv_out=None) -> None:
    """scot_gemm — see include/scot_hip.h.  compute = X3 (fp32 operands split into hi + lo while they are staged) always runs in the
    bfloat16 build: its halves keep fp32's exponent range, which the binary16 build's halves do not (gradients under the fp16 gradient
    scale overflowed there)."""
    if compute == X3 and _active != "bf16":
        prev = use("bf16")
        try:
            return gemm(layout, compute, M, N, K, A, lda, B, ldb, C, ldc, bias=bias, colscale=colscale, aux=aux, ldaux=ldaux, resid=resid,
                        ldres=ldres, a_gelu=a_gelu, b_gelu=b_gelu, accumulate=accumulate, colsum_out=colsum_out, aux_mul=aux_mul,
                        gelu_deriv_out=gelu_deriv_out)
        finally:
            use(prev)
    ws = _gemm_ws(layout, compute, M, N, K)
    rc = L().scot_gemm(layout, compute, M, N, K, ptr(A), dt(A), lda, int(a_gelu), ptr(B), dt(B), ldb, int(b_gelu),
                       ptr(C), dt(C), ldc, ptr(bias), ptr(colscale), ptr(aux), dt(aux) if aux is not None else 0, ldaux,
                       ptr(resid), dt(resid) if resid is not None else 0, ldres, int(accumulate), ptr(colsum_out),
                       ws.data_ptr(), ws.numel(),
                       int(aux_mul), ptr(gelu_deriv_out), stream())
    _lib.check(rc, "scot_gemm")


def linear_fwd(compute, x, w, out, bias=None, a_gelu=False, gelu_deriv_out=None):
    """Out[M,N] = act(x)[M,K] @ w[N,K]^T + bias;  with gelu_deriv_out: out = gelu(.), gelu_deriv_out = gelu'(.)."""
    M = x.numel() // x.shape[-1]
    N, Kw = w.shape[0], w.numel() // w.shape[0]
    gemm(NT, compute, M, N, Kw, x, x.shape[-1], w, Kw, out, out.shape[-1], bias=bias, a_gelu=a_gelu, gelu_deriv_out=gelu_deriv_out)


def linear_dgrad(compute, dy, w, dx, accumulate=False, aux=None, aux_mul=False, resid=None, wt=None):
    """dx[M,K] (+)= dy[M,N] @ w[N,K]  (* gelu'(aux), or * aux when aux_mul);  resid: dx = resid + ... (out of place).
    wt: w^T [K, N] (see transpose_cast) — the same product as the forward's NT GEMM, whose operands are both contiguous along the
    reduction index."""
    M = dy.numel() // dy.shape[-1]
    N, K = w.shape[0], w.numel() // w.shape[0]
    kw = dict(aux=aux, ldaux=aux.shape[-1] if aux is not None else 0, accumulate=accumulate, aux_mul=aux_mul, resid=resid,
              ldres=resid.shape[-1] if resid is not None else 0)
    if wt is not None:
        gemm(NT, compute, M, K, N, dy, dy.shape[-1], wt, N, dx, dx.shape[-1], **kw)
    else:
        gemm(NN, compute, M, K, N, dy, dy.shape[-1], w, K, dx, dx.shape[-1], **kw)


def transpose_cast(w, wt16, desc, n: int, tiles: int):
    """wt16 (operand format) <- per-matrix transposes of the fp32 arena w; desc int32 [n, 4] on the device (offset, rows, cols,
    first tile) — scot_transpose_cast."""
    _lib.check(L().scot_transpose_cast(ptr(w), ptr(wt16), ptr(desc), n, tiles, stream()), "scot_transpose_cast")


def linear_wgrad(compute, dy, x, dw, b_gelu=False, dbias=None):
    """dw[N,K] += dy[M,N]^T @ act(x)[M,K];  dbias[N] += Σ_m dy[m,:] (from the same LDS tiles)."""
    M = dy.numel() // dy.shape[-1]
    N, K = dy.shape[-1], x.shape[-1]
    gemm(TN, compute, N, K, M, dy, N, x, K, dw, K, b_gelu=b_gelu, accumulate=True, colsum_out=dbias)


GRAD_ADD, GRAD_STORE_SCALED, GRAD_ADD_SCALED = 0, 1, 2      # how a weight gradient meets the arena (include/scot_hip.h, scot_wgrad_group)


def wgrad_group(compute, problems, modes=None, grad_scale=None) -> bool:
    """The weight gradients of one ScOTLayer in ONE launch (+ one grouped split-K reduce): problems = [(dy, x, dw, dbias)], every
    dy [K, M_i] / x [K, N_i] in the 16-bit operand format over the SAME K rows, dw [M_i, N_i] fp32 (+=), dbias [M_i] or None.
    modes (one GRAD_* per problem, default all GRAD_ADD) / grad_scale (1-element fp32 device tensor): see the header.
    False = not covered (the caller launches them one by one)."""
    import ctypes
    n = len(problems)
    K = problems[0][0].numel() // problems[0][0].shape[-1]
    VP, IA = ctypes.c_void_p * n, ctypes.c_int * n
    dys, xs, dws = VP(*[ptr(p[0]) for p in problems]), VP(*[ptr(p[1]) for p in problems]), VP(*[ptr(p[2]) for p in problems])
    dbs = VP(*[ptr(p[3]) for p in problems])
    Ms, Ns = IA(*[p[0].shape[-1] for p in problems]), IA(*[p[1].shape[-1] for p in problems])
    for dy, x, dw, _ in problems:
        if dt(dy) != BF16 or dt(x) != BF16 or dw.dtype != torch.float32 or dy.numel() // dy.shape[-1] != K or x.numel() // x.shape[-1] != K:
            return False
    ws = workspace(int(_raw().scot_wgrad_group_workspace_bytes(n, K, Ms, Ns)))
    md = IA(*[int(m) for m in modes]) if modes is not None else None
    rc = L().scot_wgrad_group(compute, n, K, dys, xs, dws, dbs, Ms, Ns, ws.data_ptr(), ws.numel(), md, ptr(grad_scale), stream())
    if rc == -3:
        return False
    _lib.check(rc, "scot_wgrad_group")
    return True


def colsum(x, out, y=None):
    M = x.numel() // x.shape[-1]
    N = x.shape[-1]
    _lib.check(L().scot_colsum(ptr(x), dt(x), ptr(y), dt(y) if y is not None else 0, ptr(out), M, N, N, stream()), "scot_colsum")


def window_attn_fwd(compute, qkv, out, lse, bias_table, logit_scale, batch, Hp, Wp, C, heads, ws, shift):
    _lib.check(L().scot_window_attn_fwd(compute, ptr(qkv), ptr(out), ptr(lse), ptr(bias_table), ptr(logit_scale), batch, Hp, Wp,
                                        C, heads, ws, shift, stream()), "scot_window_attn_fwd")


def window_attn_probs(qkv, lse, bias_table, logit_scale, probs, batch, Hp, Wp, C, heads, ws, shift):
    """probs [batch·nW, heads, N, N] fp32 <- the block's attention probabilities, from qkv and the forward's lse (output_attentions)."""
    _lib.check(L().scot_window_attn_probs(ptr(qkv), dt(qkv), ptr(lse), ptr(bias_table), ptr(logit_scale), ptr(probs), batch, Hp, Wp, C,
                                          heads, ws, shift, stream()), "scot_window_attn_probs")


def window_attn_bwd(compute, qkv, out_fwd, dout, lse, bias_table, logit_scale, dqkv, dbias_table, dlogit_scale, batch, Hp, Wp, C,
                    heads, ws, shift):
    _lib.check(L().scot_window_attn_bwd(compute, ptr(qkv), ptr(out_fwd), ptr(dout), ptr(lse), ptr(bias_table), ptr(logit_scale), ptr(dqkv),
                                        ptr(dbias_table), ptr(dlogit_scale), batch, Hp, Wp, C, heads, ws, shift, stream()),
               "scot_window_attn_bwd")


def window_attn_bwd_rep(compute, qkv, out_fwd, dout, lse, bias_table, logit_scale, dqkv, dbias_table, dlogit_scale, batch, Hp, Wp, C,
                        heads, ws, shift, nrep, stride_tab, stride_ls):
    """window_attn_bwd with the two atomically accumulated buffers as `nrep` replicas (element strides): window w adds into replica w % nrep."""
    _lib.check(L().scot_window_attn_bwd_rep(compute, ptr(qkv), ptr(out_fwd), ptr(dout), ptr(lse), ptr(bias_table), ptr(logit_scale), ptr(dqkv),
                                            ptr(dbias_table), ptr(dlogit_scale), batch, Hp, Wp, C, heads, ws, shift, nrep, stride_tab,
                                            stride_ls, stream()), "scot_window_attn_bwd_rep")


def replica_reduce(rep, r0, nrep, stride, desc, n, max_count, dst):
    """dst[dst_off + j] += Σ_{r0 <= r < nrep} rep[r·stride + src_off + j], j < count, per int32 entry (src_off, dst_off, count) of desc."""
    _lib.check(L().scot_replica_reduce(ptr(rep), r0, nrep, stride, ptr(desc), n, max_count, ptr(dst), stream()), "scot_replica_reduce")


def cpb_fwd(coords, w0, b0, w2, table, z, ws, heads):
    _lib.check(L().scot_cpb_fwd(ptr(coords), ptr(w0), ptr(b0), ptr(w2), ptr(table), ptr(z), ws, heads, stream()), "scot_cpb_fwd")


def cpb_bwd(coords, w0, b0, w2, z, dtable, dw0, db0, dw2, ws, heads):
    _lib.check(L().scot_cpb_bwd(ptr(coords), ptr(w0), ptr(b0), ptr(w2), ptr(z), ptr(dtable), ptr(dw0), ptr(db0), ptr(dw2), ws,
                                heads, stream()), "scot_cpb_bwd")


def cpb_fwd_batched(params, desc, nlayers, max_ws, coords, tables, zbuf):
    _lib.check(L().scot_cpb_fwd_batched(ptr(params), ptr(desc), nlayers, max_ws, ptr(coords), ptr(tables), ptr(zbuf), stream()),
               "scot_cpb_fwd_batched")


def cpb_bwd_batched(params, desc, first, count, max_ws, max_heads, coords, zbuf, dtables, grads):
    _lib.check(L().scot_cpb_bwd_batched(ptr(params), ptr(desc), first, count, max_ws, max_heads, ptr(coords), ptr(zbuf), ptr(dtables),
                                        ptr(grads), stream()), "scot_cpb_bwd_batched")


def cln_fwd(x, resid, out, mean, rstd, time, gw_w, gw_b, bw_w, bw_b, rows, rows_per_sample, C, eps, out2=None, sample_scale=None):
    _lib.check(L().scot_cln_fwd(ptr(x), dt(x), ptr(resid), dt(resid) if resid is not None else 0, ptr(out), dt(out), ptr(out2),
                                dt(out2) if out2 is not None else 0, ptr(mean),
                                ptr(rstd), ptr(time), ptr(gw_w), ptr(gw_b), ptr(bw_w), ptr(bw_b), rows, rows_per_sample, C,
                                float(eps), ptr(sample_scale), stream()), "scot_cln_fwd")


def mlp_block_fwd(h16, h, w1, b1, w2, b2, out, out16, act, dact, z, mean, rstd, time, gw_w, gw_b, bw_w, bw_b, sample_scale,
                  rows, rows_per_sample, C, hid, eps) -> bool:
    """Fused fc1 → GELU → fc2 → cond-LN → residual (csrc/mlp_fused.hip).  False = shape not covered (the
    caller runs the three-kernel path); any other failure raises."""
    rc = L().scot_mlp_block_fwd(ptr(h16), ptr(h), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(out), ptr(out16), ptr(act), ptr(dact),
                                ptr(z), ptr(mean), ptr(rstd), ptr(time), ptr(gw_w), ptr(gw_b), ptr(bw_w), ptr(bw_b),
                                ptr(sample_scale), rows, rows_per_sample, C, hid, float(eps), stream())
    if rc == -3:   # SCOT_ERR_UNSUPPORTED
        return False
    _lib.check(rc, "scot_mlp_block_fwd")
    return True


def mlp_block_bwd(g, g_out, z, mean, rstd, time, gw_w, gw_b, sample_scale, dact, w1, w2, dz, du, d_gw_w, d_gw_b, d_bw_w, d_bw_b,
                  rows, rows_per_sample, C, hid) -> bool:
    """Fused cond-LN backward → dgrad fc2 (·gelu') → dgrad fc1 (+ g) (csrc/mlp_fused.hip).  False = not covered."""
    rc = L().scot_mlp_block_bwd(ptr(g), ptr(g_out), ptr(z), ptr(mean), ptr(rstd), ptr(time), ptr(gw_w), ptr(gw_b),
                                ptr(sample_scale), ptr(dact), ptr(w1), ptr(w2), ptr(dz), ptr(du), ptr(d_gw_w), ptr(d_gw_b),
                                ptr(d_bw_w), ptr(d_bw_b), rows, rows_per_sample, C, hid, stream())
    if rc == -3:   # SCOT_ERR_UNSUPPORTED
        return False
    _lib.check(rc, "scot_mlp_block_bwd")
    return True


def block_tail_fwd(proj, mlp, time, rows, rows_per_sample, C, hid, eps, wqkv=None, bqkv=None, qkv=None, z16: bool = False) -> bool:
    """The tail of a ScOTLayer's forward in one launch: proj_cln_fwd then mlp_block_fwd on its output rows (handed over through
    LDS).  proj = (attn, wo, bo, x, h, h16, z1, mean1, rstd1, gw_w1, gw_b1, bw_w1, bw_b1, sscale1); mlp = (w1, b1, w2, b2, out,
    out16, act, dact, z2, mean2, rstd2, gw_w2, gw_b2, bw_w2, bw_b2, sscale2).  wqkv [3C, C] / bqkv [3C] / qkv [rows, 3C] (optional):
    epilogue qkv = out16 · wqkv^T + bqkv — the NEXT layer's fused q/k/v projection on the rows just produced.  z16: z1 / z2 are
    16-bit tensors (only the norm backward's x-hat reads them).  act / dact None with z / statistics given = training without the
    4C-wide saves (block_tail_bwd recomputes).  False = not covered."""
    rc = L().scot_block_tail_fwd(*[ptr(t) for t in proj], *[ptr(t) for t in mlp], ptr(wqkv), ptr(bqkv), ptr(qkv), BF16 if z16 else F32,
                                 ptr(time), rows, rows_per_sample, C, hid, float(eps), stream())
    if rc == -3:
        return False
    _lib.check(rc, "scot_block_tail_fwd")
    return True


def block_tail_bwd(g, g_out, mlp, proj, time, rows, rows_per_sample, C, hid, dqkv=None, wqkv=None, h16=None, b1=None, z16: bool = False,
                   partial2=None, partial1=None) -> bool:
    """The tail of a ScOTLayer's backward in one launch: mlp_block_bwd then proj_cln_bwd on its result (which stays in
    registers in between).  mlp = (z2, mean2, rstd2, gw_w2, gw_b2, sscale2, dact, w1, w2, dz2, du, d_gw_w2, d_gw_b2, d_bw_w2,
    d_bw_b2); proj = (z1, mean1, rstd1, gw_w1, gw_b1, sscale1, wo, dz1, da, d_gw_w1, d_gw_b1, d_bw_w1, d_bw_b1).  dqkv [rows, 3C] /
    wqkv [3C, C] (optional): prologue g += dqkv · wqkv — the qkv dgrad of the layer processed before, in place (g_out is g).
    dact None: gelu'(u) is recomputed from h16 / b1; du None: not stored; z16: z1 / z2 are 16-bit; partial2 / partial1
    ([tail_workgroups, 4C | 2C] fp32): the norms' parameter-gradient column sums per workgroup instead of atomics (partial_colsum
    finishes them).  False = not covered (the caller launches the kernels one by one)."""
    rc = L().scot_block_tail_bwd(ptr(g), ptr(g_out), *[ptr(t) for t in mlp], *[ptr(t) for t in proj], ptr(dqkv), ptr(wqkv), ptr(h16), ptr(b1),
                                 BF16 if z16 else F32, ptr(partial2), ptr(partial1), ptr(time), rows, rows_per_sample, C, hid, stream())
    if rc == -3:
        return False
    _lib.check(rc, "scot_block_tail_bwd")
    return True


def memset_async(t, byte: int = 0):
    """t's bytes <- byte on the current stream (scot_memset_async: a tape entry instead of a torch call between launches)"""
    _lib.check(L().scot_memset_async(ptr(t), int(byte), t.numel() * t.element_size(), stream()), "scot_memset_async")


def memcpy_async(dst, src):
    """dst <- src (same dtype, contiguous, same device) on the current stream"""
    if dst.dtype != src.dtype or dst.numel() != src.numel() or not dst.is_contiguous() or not src.is_contiguous():
        raise TypeError("memcpy_async: contiguous tensors of one dtype and size")
    _lib.check(L().scot_memcpy_async(ptr(dst), ptr(src), dst.numel() * dst.element_size(), stream()), "scot_memcpy_async")


def event_record(event_handle: int, stream_handle):
    _lib.check(L().scot_event_record(event_handle, stream_handle), "scot_event_record")


def stream_wait_event(stream_handle, event_handle: int):
    _lib.check(L().scot_stream_wait_event(stream_handle, event_handle), "scot_stream_wait_event")


def compile_tape(cmds):
    """A recorded list of (ctypes function, arguments) / (python callable, None) -> segments: ("c", program buffer, words, kept objects)
    for every run of C-ABI calls (scot_tape_replay's word format, include/scot_hip.h), ("py", callable) for a host-side step."""
    import ctypes
    import struct
    segs, words, keep = [], [], []

    def flush():
        if words:
            buf = (ctypes.c_uint64 * len(words))(*words)
            segs.append(("c", buf, len(words), list(keep)))
            words.clear()
            keep.clear()
    for fn, args in cmds:
        if args is None:
            flush()
            segs.append(("py", fn))
            continue
        ints, flts = [], []
        for a, t in zip(args, fn.argtypes):
            if t is ctypes.c_float:
                flts.append(struct.unpack("<I", struct.pack("<f", float(a)))[0])
            elif a is None:
                ints.append(0)
            elif isinstance(a, int):
                ints.append(a & 0xFFFFFFFFFFFFFFFF)
            elif isinstance(a, ctypes.c_void_p):
                ints.append(a.value or 0)
            elif isinstance(a, ctypes.Array):
                ints.append(ctypes.addressof(a))
                keep.append(a)
            else:
                raise TypeError(f"compile_tape: argument {a!r} of {fn.__name__}")
        if len(args) != len(fn.argtypes) or len(ints) > 48 or len(flts) > 8:
            raise TypeError(f"compile_tape: {fn.__name__} does not fit the replay's calling convention")
        words += [ctypes.cast(fn, ctypes.c_void_p).value, len(ints), len(flts)] + ints + flts
    flush()
    return segs


def replay_tape(segs):
    """issue a compiled tape (the library handle is the ACTIVE build's: the entry points inside the program carry their own addresses)"""
    import ctypes
    lib = _raw()
    fail = ctypes.c_int(-1)
    for seg in segs:
        if seg[0] == "py":
            seg[1]()
        else:
            rc = lib.scot_tape_replay(seg[1], seg[2], ctypes.byref(fail))
            if rc:
                raise _lib.ScotLibraryError(f"step tape: entry {fail.value} of a replayed run returned {rc}")


def tail_workgroups(rows, rows_per_sample, C) -> int:
    """workgroups of block_tail_fwd / _bwd at these dimensions = rows of the backward's partial-sum matrices (0: not covered)"""
    return int(_raw().scot_block_tail_workgroups(rows, rows_per_sample, C))


def partial_colsum(partial, nblk, ncol, out):
    """out[j] += Σ_b partial[b][j]"""
    _lib.check(L().scot_partial_colsum(ptr(partial), nblk, ncol, ptr(out), stream()), "scot_partial_colsum")


def partial_colsum_batch(items):
    """items = [(partial, nblk, ncol, out)], at most 32: out_i[j] += Σ_b partial_i[b][j], one launch"""
    import ctypes
    n = len(items)
    VP, IA = ctypes.c_void_p * n, ctypes.c_int * n
    _lib.check(L().scot_partial_colsum_batch(n, VP(*[ptr(i[0]) for i in items]), IA(*[i[1] for i in items]), IA(*[i[2] for i in items]),
                                             VP(*[ptr(i[3]) for i in items]), stream()), "scot_partial_colsum_batch")


def wgrad_mlp(h16, dz, w1, b1, w2t, dW1, db1, dW2, db2, mode=0, grad_scale=None) -> bool:
    """fc1 / fc2 weight + bias gradients of a ScOTLayer's MLP with gelu(u), gelu'(u), du recomputed on the fly (csrc/wgrad_mlp.hip).
    dW1 | db1 | dW2 | db2 contiguous (the gradient arena's layout).  mode / grad_scale: how dW1 / dW2 meet the arena (GRAD_*).
    False = not covered."""
    M, C = h16.numel() // h16.shape[-1], h16.shape[-1]
    hid = w1.shape[0]
    need = int(_raw().scot_wgrad_mlp_workspace_bytes(M, C, hid))
    if need == 0:
        return False
    ws = workspace(need)
    rc = L().scot_wgrad_mlp(ptr(h16), ptr(dz), ptr(w1), ptr(b1), ptr(w2t), ptr(dW1), ptr(db1), ptr(dW2), ptr(db2), M, C, hid,
                            ws.data_ptr(), ws.numel(), int(mode), ptr(grad_scale), stream())
    if rc == -3:
        return False
    _lib.check(rc, "scot_wgrad_mlp")
    return True


def proj_cln_fwd(a, w, bias, resid, out, out16, z, mean, rstd, time, gw_w, gw_b, bw_w, bw_b, sample_scale, rows, rows_per_sample, C,
                 eps) -> bool:
    """Out-projection GEMM with cond-LN + residual in its epilogue (csrc/mlp_fused.hip).  False = not covered."""
    rc = L().scot_proj_cln_fwd(ptr(a), ptr(w), ptr(bias), ptr(resid), ptr(out), ptr(out16), ptr(z), ptr(mean), ptr(rstd), ptr(time),
                               ptr(gw_w), ptr(gw_b), ptr(bw_w), ptr(bw_b), ptr(sample_scale), rows, rows_per_sample, C, float(eps),
                               stream())
    if rc == -3:
        return False
    _lib.check(rc, "scot_proj_cln_fwd")
    return True


def proj_cln_bwd(g, z, mean, rstd, time, gw_w, gw_b, sample_scale, w, dz, da, d_gw_w, d_gw_b, d_bw_w, d_bw_b, rows, rows_per_sample,
                 C) -> bool:
    """Cond-LN backward → dgrad of the out-projection in one launch.  False = not covered."""
    rc = L().scot_proj_cln_bwd(ptr(g), ptr(z), ptr(mean), ptr(rstd), ptr(time), ptr(gw_w), ptr(gw_b), ptr(sample_scale), ptr(w),
                               ptr(dz), ptr(da), ptr(d_gw_w), ptr(d_gw_b), ptr(d_bw_w), ptr(d_bw_b), rows, rows_per_sample, C, stream())
    if rc == -3:
        return False
    _lib.check(rc, "scot_proj_cln_bwd")
    return True


def cln_bwd(dout, x, mean, rstd, time, gw_w, gw_b, dx, d_gw_w, d_gw_b, d_bw_w, d_bw_b, rows, rows_per_sample, C, d_xbias=None,
            sample_scale=None, mode=0, partial=None):
    """mode 0: dx and parameter gradients; 1: dx only; 2: parameter gradients only (dx may be None); 3: dx + per-block partial sums
    of the parameter gradients into `partial` (fp32, cln_bwd_partial_floats long), finished by cln_bwd_finish."""
    if mode == 3:
        wsp, wsn = partial.data_ptr(), partial.numel() * partial.element_size()
    else:
        wsp, wsn = None, 0
    _lib.check(L().scot_cln_bwd(ptr(dout), dt(dout), ptr(x), dt(x), ptr(mean), ptr(rstd), ptr(time), ptr(gw_w), ptr(gw_b),
                                ptr(dx), dt(dx) if dx is not None else 0, ptr(d_gw_w), ptr(d_gw_b), ptr(d_bw_w), ptr(d_bw_b), ptr(d_xbias), rows,
                                rows_per_sample, C, wsp, wsn, ptr(sample_scale), int(mode),
                                stream()), "scot_cln_bwd")


def cln_bwd_partial_floats(rows, rows_per_sample, C, conditional) -> int:
    """floats of scratch mode 3 of cln_bwd writes for these dimensions (0: the form does not apply)"""
    return int(_raw().scot_cln_bwd_workspace_bytes(rows, rows_per_sample, C, int(bool(conditional)))) // 4


def cln_bwd_finish(partial, rows, rows_per_sample, C, d_gw_w, d_gw_b, d_bw_w, d_bw_b):
    """parameter gradients += the per-block partial sums of a mode-3 cln_bwd"""
    _lib.check(L().scot_cln_bwd_finish(ptr(partial), rows, rows_per_sample, C, ptr(d_gw_w), ptr(d_gw_b), ptr(d_bw_w), ptr(d_bw_b), stream()),
               "scot_cln_bwd_finish")


def add(a, b, out, period=None):
    n = a.numel()
    _lib.check(L().scot_add(ptr(a), dt(a), ptr(b), dt(b), ptr(out), dt(out), n, period if period is not None else n, stream()),
               "scot_add")


def gather_pairs(data, it, src, a, b, pv, lab, T, nsrc, H, W, transpose):
    """Batch assembly from HBM-resident trajectories (poseidon_amd/data.py): data [n, T, nsrc, H, W] fp32, it int32 [3, B] =
    (trajectory, t1, t2), src int32 [C] (-1: constant plane), a / b fp32 [C]; writes pv / lab [B, C, H, W]."""
    B, C = pv.shape[0], pv.shape[1]
    _lib.check(L().scot_gather_pairs(ptr(data), ptr(it), ptr(src), ptr(a), ptr(b), ptr(pv), ptr(lab), B, C, T, nsrc, H, W,
                                     int(transpose), stream()), "scot_gather_pairs")


def pow2_rescale(v, out2):
    """out2[0] = c = 2^k >= 1 with max|v|·c in (1/2, 1], out2[1] = 1/c (device-side; see csrc/misc.hip)."""
    _lib.check(L().scot_pow2_rescale(ptr(v), v.numel(), ptr(out2), stream()), "scot_pow2_rescale")


def colscale_dev(g, gamma, mul, out, rows, C):
    """out[r, c] = g[r, c] * gamma[c] * mul[0]  (mul: device scalar)."""
    _lib.check(L().scot_colscale_dev(ptr(g), ptr(gamma), ptr(mul), ptr(out), dt(out), rows, C, stream()), "scot_colscale_dev")


def axpy_dev(dst, src, alpha, clear_src=False):
    """dst += alpha[0] * src (fp32; alpha: device scalar); clear_src: src zeroed in the same pass."""
    _lib.check(L().scot_axpy_dev(ptr(dst), ptr(src), dst.numel(), ptr(alpha), int(clear_src), stream()), "scot_axpy_dev")


def gather_planes(data, traj, tidx, src, a, b, planes, out, T, nsrc, H, W, transpose):
    """One tensor of a batch from HBM-resident trajectories with its own recipe (poseidon_amd/data.py): traj / tidx int32 [B], src int32
    [C] (-1: constant plane, -2 - p: fixed plane p of `planes` [P, H, W]), a / b fp32 [C]; writes out [B, C, H, W]."""
    B, C = out.shape[0], out.shape[1]
    _lib.check(L().scot_gather_planes(ptr(data), ptr(traj), ptr(tidx), ptr(src), ptr(a), ptr(b), ptr(planes), ptr(out), B, C, T, nsrc,
                                      H, W, int(transpose), stream()), "scot_gather_planes")


def mask_tokens(x, mask_u8, token, rows, C):
    """x[r, :] = token where mask_u8[r] (in place; reference model.py:353-359)."""
    _lib.check(L().scot_mask_tokens(ptr(x), ptr(mask_u8), ptr(token), rows, C, stream()), "scot_mask_tokens")


def mask_tokens_bwd(g, mask_u8, d_token, rows, C):
    """d_token += Σ_r mask[r]·g[r, :];  g[r, :] = 0 where mask[r]  (in place)."""
    _lib.check(L().scot_mask_tokens_bwd(ptr(g), ptr(mask_u8), ptr(d_token), rows, C, stream()), "scot_mask_tokens_bwd")


def batch_sum(x, out, batch, period):
    _lib.check(L().scot_batch_sum(ptr(x), dt(x), ptr(out), batch, period, stream()), "scot_batch_sum")


def copy2d(src, dst, B, Hs, Ws, Hd, Wd, C):
    _lib.check(L().scot_copy2d(ptr(src), dt(src), ptr(dst), dt(dst), B, Hs, Ws, Hd, Wd, C, stream()), "scot_copy2d")


def space_to_depth(fine, fine2, coarse, B, H, W, C, order):
    _lib.check(L().scot_space_to_depth(ptr(fine), ptr(fine2), dt(fine), ptr(coarse), dt(coarse), B, H, W, C, order, stream()),
               "scot_space_to_depth")


def depth_to_space(coarse, fine, B, H, W, H2, W2, C, order):
    _lib.check(L().scot_depth_to_space(ptr(coarse), dt(coarse), ptr(fine), dt(fine), B, H, W, H2, W2, C, order, stream()),
               "scot_depth_to_space")


def patchify(img, cols, B, Cc, H, W, p):
    _lib.check(L().scot_patchify(ptr(img), ptr(cols), dt(cols), B, Cc, H, W, p, stream()), "scot_patchify")


def unpatchify(cols, bias, img, B, Cc, H, W, gh, gw, p):
    _lib.check(L().scot_unpatchify(ptr(cols), dt(cols), ptr(bias), ptr(img), B, Cc, H, W, gh, gw, p, stream()), "scot_unpatchify")


def nchw_channel_sum(x, out, B, Cc, HW):
    _lib.check(L().scot_nchw_channel_sum(ptr(x), ptr(out), B, Cc, HW, stream()), "scot_nchw_channel_sum")


def cast(src, dst):
    """dst <- src (dtype conversion, one pass) — scot_scale_residual with no scale / residual."""
    n = src.numel()
    _lib.check(L().scot_scale_residual(ptr(src), dt(src), None, None, 0, ptr(dst), dt(dst), 1, n, stream()), "scot_scale_residual(cast)")


def spectral_apply(U, Pr, Pi, Y, nimg: int, s: int, t: int):
    """Y[b] = Pr · U[b, :, :t] - Pi · U[b, :, t:]  (U: [nimg, s, 2t], Pr/Pi: [t, s], Y: [nimg, t, t]; all fp32)."""
    _lib.check(L().scot_spectral_apply(ptr(U), ptr(Pr), ptr(Pi), ptr(Y), nimg, s, t, stream()), "scot_spectral_apply")


def dp_pack(src, wire, scale: float):
    """wire (bfloat16) <- scale * src (fp32): the gradient arena's wire format for the data-parallel exchange, one pass."""
    if src.dtype != torch.float32 or wire.dtype != torch.bfloat16 or wire.numel() < src.numel():
        raise TypeError("dp_pack: fp32 source, bfloat16 wire buffer of at least the same length")
    _lib.check(L().scot_dp_pack(ptr(src), ptr(wire), src.numel(), float(scale), stream()), "scot_dp_pack")


def dp_unpack(wire, dst, scale: float = 1.0):
    """dst (fp32) <- scale * wire (bfloat16)."""
    if dst.dtype != torch.float32 or wire.dtype != torch.bfloat16 or wire.numel() < dst.numel():
        raise TypeError("dp_unpack: bfloat16 wire buffer, fp32 destination")
    _lib.check(L().scot_dp_unpack(ptr(wire), ptr(dst), dst.numel(), float(scale), stream()), "scot_dp_unpack")


# --- the exchange behind the C ABI (csrc/dp.hip): one RCCL communicator per process, owned by ONE build of the library (the
# communicator is library state and bf16 / f16 are two libraries; the wire format is bfloat16 in both) ---
def _dp_lib():
    return _lib.load(kind="bf16")


def dp_unique_id() -> bytes:
    """The 128-byte rendezvous token of a new communicator (drawn by rank 0, handed to every rank by the host)."""
    import ctypes
    buf = ctypes.create_string_buffer(128)
    _lib.check(_dp_lib().scot_dp_unique_id(ctypes.cast(buf, ctypes.c_void_p)), "scot_dp_unique_id")
    return buf.raw


def dp_init(unique_id: bytes, rank: int, world: int):
    """Collective: join the communicator on the current HIP device."""
    import ctypes
    if len(unique_id) != 128:
        raise ValueError("dp_init: the unique id is 128 bytes (ops.dp_unique_id() on rank 0)")
    buf = ctypes.create_string_buffer(bytes(unique_id), 128)
    _lib.check(_dp_lib().scot_dp_init(ctypes.cast(buf, ctypes.c_void_p), int(rank), int(world)), "scot_dp_init")


def dp_allreduce(buf):
    """In-place SUM over the native communicator's ranks of a flat fp32 or bfloat16 device tensor, on the current stream."""
    if buf.dtype not in (torch.float32, torch.bfloat16) or not buf.is_contiguous():
        raise TypeError("dp_allreduce: contiguous fp32 or bfloat16 (wire format) tensor")
    _lib.check(_dp_lib().scot_dp_allreduce_bucket(ptr(buf), buf.numel(), 0 if buf.dtype == torch.float32 else 1, stream()),
               "scot_dp_allreduce_bucket")


def dp_world() -> int:
    return int(_dp_lib().scot_dp_world())


def dp_rank() -> int:
    return int(_dp_lib().scot_dp_rank())


def dp_finalize():
    _lib.check(_dp_lib().scot_dp_finalize(), "scot_dp_finalize")


def scale_inplace(x, scale: float, nonfinite=None):
    """x (flat fp32, 16-byte aligned) *= scale; nonfinite (int32[1], optional) counts waves that saw Inf/NaN."""
    _lib.check(L().scot_scale_inplace(ptr(x), x.numel(), float(scale), ptr(nonfinite), stream()), "scot_scale_inplace")


def segments_scale(x, chunks, nchunks: int, scale_dev=None, nonfinite=None):
    """the pieces `chunks` (int64 [nchunks, 2] on the device: offset, count in floats) of the flat fp32 tensor x are multiplied by
    scale_dev[0], or zeroed when scale_dev is None — one launch (scot_segments_scale)"""
    if nchunks:
        _lib.check(L().scot_segments_scale(ptr(x), ptr(chunks), int(nchunks), ptr(scale_dev), ptr(nonfinite), stream()), "scot_segments_scale")


def scale_inplace_dev(x, scale_dev, nonfinite=None):
    """x *= scale_dev[0] (a 1-element fp32 device tensor: the fp16 build's dynamic gradient scale or its reciprocal)"""
    _lib.check(L().scot_scale_inplace_dev(ptr(x), x.numel(), ptr(scale_dev), ptr(nonfinite), stream()), "scot_scale_inplace_dev")


def scale_residual(y, scale, resid, out, rows, N):
    _lib.check(L().scot_scale_residual(ptr(y), dt(y), ptr(scale), ptr(resid), dt(resid) if resid is not None else 0, ptr(out),
                                       dt(out), rows, N, stream()), "scot_scale_residual")


def dwconv7(x, w, bias, y, B, H, W, C, flip=False):
    _lib.check(L().scot_dwconv7(ptr(x), dt(x), ptr(w), ptr(bias), ptr(y), dt(y), B, H, W, C, int(flip), stream()), "scot_dwconv7")


def dwconv7_wgrad(dy, x, dw, db, B, H, W, C):
    _lib.check(L().scot_dwconv7_wgrad(ptr(dy), dt(dy), ptr(x), dt(x), ptr(dw), ptr(db), B, H, W, C, stream()), "scot_dwconv7_wgrad")


def conv5(inp, w, out, B, Cc, H, W, transpose=False):
    _lib.check(L().scot_conv5(ptr(inp), ptr(w), ptr(out), B, Cc, H, W, int(transpose), stream()), "scot_conv5")


def conv5_wgrad(dout, inp, dw, B, Cc, H, W):
    _lib.check(L().scot_conv5_wgrad(ptr(dout), ptr(inp), ptr(dw), B, Cc, H, W, stream()), "scot_conv5_wgrad")


def head_finalize(pred, pv, pv_ch, labels, mask, mask_full, group_of_channel, sums, B, Cc, HW, p):
    _lib.check(L().scot_head_finalize(ptr(pred), ptr(pv), pv_ch, ptr(labels), ptr(mask), int(mask_full), ptr(group_of_channel),
                                      ptr(sums), B, Cc, HW, p, stream()), "scot_head_finalize")


def loss_finish(sums, counts, G, normalized, loss):
    _lib.check(L().scot_loss_finish(ptr(sums), ptr(counts), G, int(normalized), ptr(loss), stream()), "scot_loss_finish")


def loss_bwd(pred, labels, mask, mask_full, group_of_channel, sums, counts, G, normalized, dloss, dpred, B, Cc, HW, p):
    _lib.check(L().scot_loss_bwd(ptr(pred), ptr(labels), ptr(mask), int(mask_full), ptr(group_of_channel), ptr(sums), ptr(counts),
                                 G, int(normalized), ptr(dloss), ptr(dpred), B, Cc, HW, p, stream()), "scot_loss_bwd")
