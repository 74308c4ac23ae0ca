"""Which GEMM-operand formats meet the north star's 1e-3 (CPU experiment, no GPU needed).

Runs the CPU oracle's forward on a golden fixture's configuration with every ScOTLayer / ConvNeXt GEMM operand rounded to a
chosen format before an exact (fp32-accumulated) product — i.e. what an MFMA with fp32 accumulation does to the value — and
prints the output's rel-L2 against the fixture of the real reference.  The "trunk" (patch embed / merge / unmerge / recovery)
stays fp32, as in the engine.  Formats: bf16 (8-bit mantissa), fp16 (11), `x` = exact (what a hi+lo split delivers).
  python tools/probes/precision_sim.py [fixture] -- acts/weights/attention formats per row below.
Test infrastructure only (imports oracle/)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_fixture, rel_l2  # noqa: E402
from oracle import scot_cpu  # noqa: E402
from poseidon_amd.config import ScOTConfig  # noqa: E402
from poseidon_amd.geometry import param_shapes  # noqa: E402
from poseidon_amd.synth import synth_inputs, synth_state_dict  # noqa: E402

R = {"x": lambda t: t, "bf16": lambda t: t.to(torch.bfloat16).to(torch.float32), "fp16": lambda t: t.to(torch.float16).to(torch.float32)}
MODE = {"act": "x", "w": "x", "attn": "x", "trunk": False, "site": None, "n": 0, "per_site": {}}
_orig = torch.Tensor.__matmul__


def _mm(a, b):
    if MODE["trunk"] and a.dim() >= 3 and MODE.get("trunk_fmt"):      # patch embed / merge / unmerge / recovery GEMMs
        return _orig(R[MODE["trunk_fmt"]](a), R[MODE["trunk_fmt"]](b))
    if MODE["trunk"] or a.dim() < 3:
        return _orig(a, b)
    if b.dim() == 2:
        site = MODE["site"]
        if site == "attn":      # q, k, v projections then the output projection (oracle.attention's call order)
            MODE["n"] += 1
            site = "qkv" if MODE["n"] <= 3 else "proj"
        elif site == "layer":   # after attention(): fc1 then fc2
            MODE["n"] += 1
            site = "fc1" if MODE["n"] == 1 else "fc2"
        act, w = MODE["per_site"].get(site, (MODE["act"], MODE["w"]))
        return _orig(R[act](a), R[w](b))
    return _orig(R[MODE["attn"]](a), R[MODE["attn"]](b))


def _site(fn, name, after=None):
    def w(*a, **k):
        prev = (MODE["site"], MODE["n"])
        MODE["site"], MODE["n"] = name, 0
        try:
            return fn(*a, **k)
        finally:
            MODE["site"], MODE["n"] = (after, 0) if after else prev
    return w


def _trunk(fn):
    def w(*a, **k):
        MODE["trunk"] = True
        try:
            return fn(*a, **k)
        finally:
            MODE["trunk"] = False
    return w


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "poseidonT_trained"
    f, meta = load_fixture(name)
    cfg = ScOTConfig(**meta["cfg"])
    sd = synth_state_dict(param_shapes(cfg), meta["regime"])
    pv, t, lab = synth_inputs(meta["batch"], cfg.num_channels, cfg.num_out_channels, meta.get("size", cfg.image_size), meta["kind"])
    for fn in ("patch_embed", "patch_merge", "patch_unmerge", "patch_recovery"):
        setattr(scot_cpu, fn, _trunk(getattr(scot_cpu, fn)))
    scot_cpu.attention = _site(scot_cpu.attention, "attn", after="layer")
    scot_cpu.convnext = _site(scot_cpu.convnext, "cnx")
    torch.Tensor.__matmul__ = _mm
    if len(sys.argv) > 2 and sys.argv[2] == "trunk":
        with torch.no_grad():
            for tf in (None, "fp16", "bf16"):
                MODE.update(act="fp16", w="fp16", attn="fp16", per_site={}, trunk_fmt=tf)
                _, pred = scot_cpu.scot_forward(sd, cfg, pv, t if cfg.use_conditioning else None, lab)
                print(f"  fp16 everywhere, trunk (patch embed / merge / unmerge / recovery) {tf or 'fp32'}: {rel_l2(pred.numpy(), f['output']):.2e}")
        return
    if len(sys.argv) > 2 and sys.argv[2] == "sites":
        X, H = ("x", "fp16"), ("fp16", "fp16")
        cases = [("all fp16", {}), ("proj act exact", {"proj": X}), ("fc2 act exact", {"fc2": X}), ("proj+fc2 act exact", {"proj": X, "fc2": X}),
                 ("proj+fc2+fc1 act exact", {"proj": X, "fc2": X, "fc1": X}), ("proj+fc2+fc1+qkv act exact", {"proj": X, "fc2": X, "fc1": X, "qkv": X}),
                 ("all linear act exact (incl. ConvNeXt)", {"proj": X, "fc2": X, "fc1": X, "qkv": X, "cnx": X}),
                 ("ConvNeXt exact", {"cnx": ("x", "x")}), ("ConvNeXt + proj exact", {"cnx": ("x", "x"), "proj": ("x", "x")}),
                 ("ConvNeXt, proj, fc2 exact", {"cnx": ("x", "x"), "proj": ("x", "x"), "fc2": ("x", "x")}),
                 ("fc1, qkv exact", {"fc1": ("x", "x"), "qkv": ("x", "x")})]
        with torch.no_grad():
            for label, per in cases:
                MODE.update(act="fp16", w="fp16", attn="fp16", per_site=per)
                _, pred = scot_cpu.scot_forward(sd, cfg, pv, t if cfg.use_conditioning else None, lab)
                print(f"  {label:45s}: {rel_l2(pred.numpy(), f['output']):.2e}")
        return
    rows = [("x", "x", "x"), ("bf16", "bf16", "bf16"), ("fp16", "fp16", "fp16"), ("fp16", "x", "fp16"), ("x", "fp16", "fp16"),
            ("fp16", "x", "x"), ("x", "x", "fp16"), ("fp16", "fp16", "x"), ("bf16", "x", "bf16"), ("x", "x", "bf16")]
    print(f"{name}: output rel-L2 vs the reference fixture")
    with torch.no_grad():
        for act, w, attn in rows:
            MODE.update(act=act, w=w, attn=attn)
            _, pred = scot_cpu.scot_forward(sd, cfg, pv, t if cfg.use_conditioning else None, lab)
            print(f"  activations {act:5s} weights {w:5s} attention(QK^T, PV) {attn:5s}: {rel_l2(pred.numpy(), f['output']):.2e}")


if __name__ == "__main__":
    main()
