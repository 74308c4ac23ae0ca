"""Micro-benchmark of scot_deep_tail_fwd / _bwd alone (cold weights: NL different layers' worth of operands rotate).
usage: python tools/bench_deep_tail.py [C] [rows] [rows_per_sample] [hsplit]"""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd import ops

DEV = "cuda"
C = int(sys.argv[1]) if len(sys.argv) > 1 else 384
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
L = int(sys.argv[3]) if len(sys.argv) > 3 else 64
HS = int(sys.argv[4]) if len(sys.argv) > 4 else 1
NL = int(os.environ.get("NL", "24"))
ops.use("f16")
bf = ops.half_dtype()
hid = 4 * C
B = M // L


def fragpack(w32, mode=0):
    N, K = w32.shape
    desc = torch.tensor([[0, N, K, 0, mode, 0]], dtype=torch.int32, device=DEV)
    out = torch.empty(N * K, device=DEV, dtype=bf)
    ops.fragpack(w32.contiguous().view(-1), out, desc, 1, (N * K // 8 + 255) // 256)
    return out


g = torch.Generator(device="cpu").manual_seed(0)
rn = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(DEV)
layers = []
for i in range(NL):
    layers.append(dict(wo=fragpack(rn(C, C, scale=C ** -0.5)), w1=fragpack(rn(hid, C, scale=C ** -0.5), 2), w2=fragpack(rn(C, hid, scale=hid ** -0.5)),
                       wq=fragpack(rn(3 * C, C, scale=C ** -0.5))))
a, x = rn(M, C).to(bf), rn(M, C)
bo, b1, b2, bq = rn(C, scale=0.2), rn(hid, scale=0.2), rn(C, scale=0.2), rn(3 * C, scale=0.2)
t = torch.rand(B, device=DEV)
n1 = [rn(C, scale=0.3), 1 + rn(C, scale=0.1), rn(C, scale=0.1), rn(C, scale=0.1)]
n2 = [rn(C, scale=0.3), 1 + rn(C, scale=0.1), rn(C, scale=0.1), rn(C, scale=0.1)]
e = lambda *s, dtype=torch.float32: torch.empty(*s, device=DEV, dtype=dtype)


def run(train, qkv, reps=48, nout=1):
    ds = [mk(train, qkv) for _ in range(nout)]
    yp = e(HS, M, C) if HS > 1 else None

    def one(w, d):
        ops.deep_tail_fwd((a, w["wo"], bo, x, d["h"], d["h16"], d["z1"], d["m1"], d["r1"], n1[0], n1[1], n1[2], n1[3], None),
                          (w["w1"], b1, w["w2"], b2, d["out"], d["out16"], d["u"], d["gp"], d["z2"], d["m2"], d["r2"], n2[0], n2[1], n2[2], n2[3], None),
                          t, M, L, C, hid, 1e-5, *((w["wq"], bq, d["q"]) if d["q"] is not None else ()), hsplit=HS, ypart=yp)
        if HS > 1:
            ops.deep_tail_finish(yp, HS, b2, d["h"], d["out"], d["out16"], d["z2"], d["m2"], d["r2"], n2[0], n2[1], n2[2], n2[3], None, t, M, L, C, 1e-5)
    for i in range(4):
        one(layers[i % NL], ds[i % nout])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        one(layers[i % NL], ds[i % nout])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


def mk(train, qkv):
    d = dict(h=e(M, C) if HS > 1 else None, h16=e(M, C, dtype=bf) if train else None, out=e(M, C), out16=e(M, C, dtype=bf),
             q=e(M, 3 * C, dtype=bf) if (qkv and HS == 1) else None)
    d.update(dict(z1=e(M, C), m1=e(M), r1=e(M), u=e(M, hid, dtype=bf), gp=e(M, hid, dtype=bf), z2=e(M, C), m2=e(M), r2=e(M)) if train else
             dict(z1=None, m1=None, r1=None, u=None, gp=None, z2=None, m2=None, r2=None))
    return d


print(f"deep_tail_fwd C={C} rows={M} hsplit={HS}: train+qkv {run(True, True):.1f} us | train {run(True, False):.1f} | "
      f"inference+qkv {run(False, True):.1f} | inference {run(False, False):.1f} | 16 rotating output sets: train+qkv {run(True, True, nout=16):.1f} train {run(True, False, nout=16):.1f} "
      f"inference+qkv {run(False, True, nout=16):.1f} inference {run(False, False, nout=16):.1f}")
