"""Ablation study of ONE kernel source (cdna_hip_programming.md §5.4: "ablate empirically before optimizing").

  python tools/ablate_kernels.py build            (here: cross-compiles variants of csrc/mlp_fused.hip with -DSCOT_ABL=<bits> into tools/_abl/)
  python tools/ablate_kernels.py run [fwd|bwd]    (on the MI355X: times scot_block_tail_fwd / _bwd of every variant, cold operands)

SCOT_ABL bits (csrc/common.h): 1 no st8 stores, 2 trivial GELU, 4 no ld8 loads, 8 no MFMA, 16 no workgroup barriers.  Timing only: the
variants compute garbage.  The product library is never built with SCOT_ABL."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_abl")
CSRC = os.path.join(ROOT, "poseidon_amd", "csrc")
VARIANTS = [0, 1, 2, 8, 16, 1 | 2, 1 | 2 | 8, 1 | 2 | 8 | 16, 1 | 2 | 4 | 8 | 16]
KIND = os.environ.get("ABL_KIND", "f16")          # operand format of the variants (attention micro-benchmarks use bf16 tensors)
if os.environ.get("ABL_SET") == "attn":
    VARIANTS = [0, 8, 16, 256, 32, 8 | 256, 8 | 16 | 32 | 256]
if os.environ.get("ABL_SET") == "wm":
    VARIANTS = [0, 2, 8, 16, 2 | 8, 2 | 8 | 16]
if os.environ.get("ABL_SET") == "bwd2":
    VARIANTS = [0, 32, 64, 128, 64 | 128, 32 | 64 | 128, 1 | 4 | 32 | 64 | 128, 1 | 2 | 4 | 8 | 16 | 32 | 64 | 128]
if os.environ.get("ABL_VARIANTS"):          # explicit list of SCOT_ABL values, e.g. "0,8,32,5"
    VARIANTS = [int(v) for v in os.environ["ABL_VARIANTS"].split(",")]
SRC = os.environ.get("ABL_SRC", "mlp_fused.hip")


def build():
    sys.path.insert(0, ROOT)
    from poseidon_amd import build as b
    b.build()
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for v in VARIANTS:
        obj = os.path.join(OUT, f"{SRC[:-4]}_{v}.o")
        defs = ["-DSCOT_OPERAND_FP16"] if KIND == "f16" else []
        procs.append((v, obj, subprocess.Popen([b.HIPCC, *b.FLAGS, *defs, f"-DSCOT_ABL={v}", "-c", os.path.join(CSRC, SRC), "-o", obj])))
    for v, obj, p in procs:
        assert p.wait() == 0, v
        others = [os.path.join(CSRC, s.replace(".hip", ".f16.o" if KIND == "f16" else ".o")) for s in b.SOURCES if s != SRC]
        subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", obj, *others, "-o", os.path.join(OUT, f"libscot_abl_{v}.so")])
        os.remove(obj)
    print("built", len(VARIANTS), "variants in", OUT)


def worker(which):
    import torch
    sys.path.insert(0, ROOT)
    from poseidon_amd import ops
    ops.use("f16")
    hd = ops.half_dtype()
    dev = "cuda"
    B = 64
    res = []
    for L, C in [(1024, 96), (256, 192)]:
        M, hid = B * L, 4 * C
        per = M * C * 40 + M * hid * 6
        nset = max(3, int(1.5e9 / per) + 1)
        g = lambda *s_, d=torch.float32, sc=1.0: (torch.randn(*s_, device=dev) * sc).to(d)
        wo, w1, w2, wq = g(C, C, d=hd, sc=C ** -0.5), g(hid, C, d=hd, sc=C ** -0.5), g(C, hid, d=hd, sc=hid ** -0.5), g(3 * C, C, d=hd, sc=C ** -0.5)
        bo, b1, b2, bq = g(C, sc=0.1), g(hid, sc=0.1), g(C, sc=0.1), g(3 * C, sc=0.1)
        n1, n2 = [g(C) for _ in range(4)], [g(C) for _ in range(4)]
        gr = [torch.zeros(C, device=dev) for _ in range(8)]
        t = torch.rand(B, device=dev)
        sets = []
        for _ in range(nset):
            sets.append(dict(a=g(M, C, d=hd), x=g(M, C), h=torch.empty(M, C, device=dev), h16=torch.empty(M, C, device=dev, dtype=hd),
                             z1=torch.empty(M, C, device=dev), st=[torch.rand(M, device=dev) + 0.5 for _ in range(4)],
                             out=torch.empty(M, C, device=dev), out16=torch.empty(M, C, device=dev, dtype=hd),
                             act=torch.empty(M, hid, device=dev, dtype=hd), dact=g(M, hid, d=hd), z2=g(M, C), qkv=g(M, 3 * C, d=hd),
                             g=g(M, C), dz2=torch.empty(M, C, device=dev, dtype=hd), du=torch.empty(M, hid, device=dev, dtype=hd),
                             dz1=torch.empty(M, C, device=dev, dtype=hd), da=torch.empty(M, C, device=dev, dtype=hd), zz1=g(M, C)))
        it = [0]

        def fwd(train, qkv=True):
            s = sets[it[0] % nset]
            it[0] += 1
            ok = ops.block_tail_fwd((s["a"], wo, bo, s["x"], s["h"], s["h16"], s["z1"] if train else None, s["st"][0] if train else None,
                                     s["st"][1] if train else None, n1[0], n1[1], n1[2], n1[3], None),
                                    (w1, b1, w2, b2, s["out"], s["out16"], s["act"] if train == 2 else None, s["dact"] if train == 2 else None,
                                     s["z2"] if train else None, s["st"][2] if train else None, s["st"][3] if train else None, n2[0], n2[1], n2[2], n2[3], None),
                                    t, M, L, C, hid, 1e-5, *((wq, bq, s["qkv"]) if qkv else (None, None, None)))
            assert ok

        def bwd(pro):
            s = sets[it[0] % nset]
            it[0] += 1
            ok = ops.block_tail_bwd(s["g"], s["g"], (s["z2"], s["st"][2], s["st"][3], n2[0], n2[1], None, s["dact"], w1, w2, s["dz2"], s["du"], gr[0], gr[1], gr[2], gr[3]),
                                    (s["zz1"], s["st"][0], s["st"][1], n1[0], n1[1], None, wo, s["dz1"], s["da"], gr[4], gr[5], gr[6], gr[7]),
                                    t, M, L, C, hid, dqkv=s["qkv"] if pro else None, wqkv=wq if pro else None)
            assert ok

        def timeit(fn):
            for _ in range(nset):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3 * nset):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / (3 * nset) * 1e3
        if which == "fwd":
            res += [f"C={C} fwd train+act {timeit(lambda: fwd(2)):6.1f}", f"fwd train-noact {timeit(lambda: fwd(1)):6.1f}", f"fwd infer {timeit(lambda: fwd(0)):6.1f}",
                    f"fwd train+act noqkv {timeit(lambda: fwd(2, False)):6.1f}"]
        else:
            res += [f"C={C} bwd {timeit(lambda: bwd(False)):6.1f}", f"bwd+pro {timeit(lambda: bwd(C == 96)):6.1f}"]
        del sets
    print(" | ".join(res), flush=True)


def run(which):
    if which == "attn":
        names = {8: "nomfma", 16: "nobarrier", 256: "noexp", 32: "noatomic"}
        for v in VARIANTS:
            env = dict(os.environ, SCOT_LIB_BF16=os.path.join(OUT, f"libscot_abl_{v}.so"), BK_COLD="1")
            label = "+".join(n for b_, n in names.items() if v & b_) or "full"
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_kernels.py"), "attn16"], env=env, capture_output=True, text=True, timeout=300)
            lines = [l.split(":", 1)[1].strip() for l in out.stdout.splitlines() if l.startswith("attn cold")]
            print(f"{label:34s} " + " | ".join(lines) + (out.stderr[-200:] if not lines else ""), flush=True)
        return
    if which == "wm":
        for v in VARIANTS:
            env = dict(os.environ, SCOT_LIB_F16=os.path.join(OUT, f"libscot_abl_{v}.so"))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_wgrad_mlp.py")], env=env, capture_output=True, text=True, timeout=300)
            print(f"abl={v:3d} {out.stdout.strip() or out.stderr[-300:]}", flush=True)
        return
    names = {1: "nostore", 2: "nogelu", 4: "noload", 8: "nomfma", 16: "nobarrier", 32: "noatomic", 64: "nodact", 128: "nodu"}
    for v in VARIANTS:
        env = dict(os.environ, SCOT_LIB_F16=os.path.join(OUT, f"libscot_abl_{v}.so"))
        label = "+".join(n for b, n in names.items() if v & b) or "full"
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "worker", which], env=env, capture_output=True, text=True, timeout=300)
        print(f"{label:36s} {out.stdout.strip() or out.stderr[-300:]}", flush=True)


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "build"
    if cmd == "build":
        build()
    elif cmd == "worker":
        worker(sys.argv[2])
    else:
        for w in (sys.argv[2:] or ["fwd", "bwd"]):
            print("==", w)
            run(w)
