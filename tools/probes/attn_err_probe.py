import sys, math, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_kernels_gpu as T
from poseidon_amd import ops
rel, rnd, DEV = T.rel, T.rnd, T.DEV
for kind in ("bf16", "f16"):
    ops.use(kind)
    for compute in (ops.BF16,):
        for case in T.ATTN_CASES:
            B, Hp, Wp, C, heads, ws, shift = case
            cdt = ops.HALF[kind]
            L, TS, N = Hp * Wp, (2 * ws - 1) ** 2, ws * ws
            qkv = rnd(B, L, 3 * C, dtype=cdt)
            table = (16 * torch.sigmoid(rnd(heads, TS, seed=1))).contiguous()
            ls = torch.linspace(math.log(3.0), math.log(20.0), heads, device=DEV)
            dout = rnd(B, L, C, dtype=cdt, seed=2)
            out = torch.full((B, L, C), float("nan"), device=DEV, dtype=cdt)
            nW = (Hp // ws) * (Wp // ws)
            lse = torch.empty(B * nW, heads, N, device=DEV)
            ops.window_attn_fwd(compute, qkv, out, lse, table, ls, B, Hp, Wp, C, heads, ws, shift)
            dqkv = torch.full((B, L, 3 * C), float("nan"), device=DEV, dtype=cdt)
            dtab = torch.zeros(heads, TS, device=DEV); dls = torch.zeros(heads, device=DEV)
            ops.window_attn_bwd(compute, qkv, out, dout, lse, table, ls, dqkv, dtab, dls, B, Hp, Wp, C, heads, ws, shift)
            torch.cuda.synchronize()
            q64 = qkv.double().requires_grad_(True); t64 = table.double().requires_grad_(True); l64 = ls.double().requires_grad_(True)
            ref = T._attn_ref(q64, t64, l64, B, Hp, Wp, C, heads, ws, shift)
            ref.backward(dout.double())
            print(kind, case, "out %.2e dqkv %.2e dtab %.2e dls %.2e" % (rel(out, ref.detach()), rel(dqkv, q64.grad), rel(dtab, t64.grad), rel(dls, l64.grad)))
