"""Whole-model parity on the MI355X: ScOT (HIP engine) vs (a) golden vectors produced by the real reference and
(b) the CPU oracle on the same closed-form inputs.  compute='fp32' must meet the 1e-5 class bound; compute='bf16'
is reported against the north-star 1e-3 on both parameter regimes (bf16 GEMM operands alone put the
reference itself at 6e-3..2e-2 on trained-like weights, so the trained-regime bound asserted here is looser)."""
import json
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_fixture, rel_l2  # noqa: E402
from poseidon_amd.config import ScOTConfig  # noqa: E402
from poseidon_amd.geometry import param_shapes  # noqa: E402
from poseidon_amd.synth import apply_obstacle, generate_on, synth_inputs, synth_obstacle_mask, synth_state_dict  # noqa: E402
from poseidon_amd import ops  # noqa: E402
from scOT.model import ScOT  # noqa: E402

DEV = "cuda"


def build(meta, compute):
    cfg = ScOTConfig(**meta["cfg"])
    # (fixtures are batches of 1-2: fused_min_rows=0 keeps them on the fused layer tails — the kernels the timed batches select — which the
    #  engine's policy otherwise leaves to >= 4096 token rows; the layer-by-layer path at small batches is what `_preset_model` runs below)
    model = ScOT(cfg, compute=compute, engine_options={"fused_min_rows": 0})
    model.load_state_dict(synth_state_dict(param_shapes(cfg), meta["regime"]))
    return cfg, model.to(DEV)


def inputs(cfg, meta):
    size = meta.get("size", cfg.image_size)
    pv, t, lab = synth_inputs(meta["batch"], cfg.num_channels, cfg.num_out_channels, size, meta["kind"])
    pm = None
    if meta.get("with_mask") == "obstacle":  # (B,1,H,W) mask, Airfoil-style (tests/golden/make_obstacle_fixture.py)
        pm = synth_obstacle_mask(meta["batch"], size)
        pv, lab = apply_obstacle(pv, lab, pm)
    kw = dict(pixel_values=pv.to(DEV), labels=lab.to(DEV))
    if cfg.use_conditioning:
        kw["time"] = t.to(DEV)
    if pm is not None:
        kw["pixel_mask"] = pm.to(DEV)
    elif meta.get("with_mask"):
        pm = torch.zeros(meta["batch"], cfg.num_out_channels, dtype=torch.bool)
        pm[:, -1] = True
        kw["pixel_mask"] = pm.to(DEV)
    return kw


def grads_report(model, f, tol_each, tol_global, floor=1e-9, skip=()):
    num = den = 0.0
    worst = (0.0, "")
    for k, p in model.named_parameters():
        key = "grad:" + k
        if key not in f.files or any(s in k for s in skip):
            continue
        ref = f[key].astype(np.float64)
        g = p.grad.detach().cpu().numpy().astype(np.float64)
        err = float(np.linalg.norm(g - ref))
        nr = float(np.linalg.norm(ref))
        if nr > 0 and err / nr > worst[0]:
            worst = (err / nr, k)
        assert err < tol_each * nr + floor, (k, err / max(nr, 1e-30))
        num += err ** 2
        den += nr ** 2
    if den > 0:
        assert (num / den) ** 0.5 < tol_global, ((num / den) ** 0.5, worst)
    return (num / max(den, 1e-300)) ** 0.5, worst


TINY = ["tiny_trained", "tiny_hf", "tiny_odd", "tiny_shift3", "tiny_nocond_p2", "tiny_learnres_mask", "tiny_obstacle_mask", "tiny_abspos",
        "tiny_w16"]


@pytest.mark.parametrize("name", TINY)
def test_fp32_vs_reference_fixture(name):
    f, meta = load_fixture(name)
    cfg, model = build(meta, "fp32")
    out = model(**inputs(cfg, meta))
    out.loss.backward()
    torch.cuda.synchronize()
    e_out = rel_l2(out.output.detach().cpu().numpy(), f["output"])
    e_loss = abs(float(out.loss.detach()) - float(f["loss"])) / abs(float(f["loss"]))
    print(f"\n[{name} fp32] out rel-L2 {e_out:.2e}  loss rel {e_loss:.2e}")
    assert e_out < 1e-5 + 5e-6  # 1e-5 north-star bound + the fixture's own fp32 noise (tests/test_oracle_golden.py)
    assert e_loss < 2e-5
    g, worst = grads_report(model, f, tol_each=1e-3, tol_global=1e-4)
    print(f"[{name} fp32] grads global rel-L2 {g:.2e}, worst {worst}")


@pytest.mark.parametrize("name", ["tiny_trained", "tiny_hf", "tiny_shift3", "tiny_w16"])
def test_bf16_vs_reference_fixture(name):
    f, meta = load_fixture(name)
    cfg, model = build(meta, "bf16")
    out = model(**inputs(cfg, meta))
    out.loss.backward()
    torch.cuda.synchronize()
    e_out = rel_l2(out.output.detach().cpu().numpy(), f["output"])
    # per-tensor bound is vacuous in bf16 for the cancellation-dominated scalars (logit_scale, CPB-MLP): global only
    # d logit_scale = Σ_{windows,q,k} dS·cos cancels over keys (Σ_k dS = 0) AND over rows/windows: with bf16-stored qkv/dO its
    # absolute error is O(2^-9·Σ|dS·cos|) whatever the kernel does (measured up to 6x its tiny true value on tiny_shift3),
    # so it is reported but excluded from the bound; everything else is bounded globally.
    # tiny models (head_dim 16, 16-token windows, O(10) activations in the trained-like regime) amplify bf16 noise in
    # the backward far more than the presets do (Poseidon-T/B: median grad-norm deviation 4e-3..9e-3): regression guard only.
    g, worst = grads_report(model, f, tol_each=1e9, tol_global=0.7, floor=1e-6, skip=("logit_scale",))
    print(f"\n[{name} bf16] out rel-L2 {e_out:.2e}; grads global rel-L2 {g:.2e}, worst {worst}")
    # MEASURED on MI355X (round 1, fp32 trunk + bf16 branches): hf regime 2e-4 (tiny) .. 2e-3 (Poseidon-B), trained-like
    # regime 6e-3..3e-2 — the north-star 1e-3 is NOT met by single-pass bf16 operands in the branches
    # (DESIGN.md "Numerics"); the bounds below only guard regressions.
    assert e_out < (3e-3 if meta["regime"] == "hf" else 5e-2)


@pytest.mark.parametrize("name,compute", [("poseidonT_trained", "fp32"), ("poseidonT_hf", "fp32"), ("poseidonT_trained", "bf16"),
                                          ("poseidonT_hf", "bf16"), ("poseidonB_trained", "fp32"), ("poseidonB_trained", "bf16"),
                                          ("poseidonB_hf", "bf16"), ("poseidonT_trained", "bf16x3"), ("poseidonT_hf", "bf16x3"),
                                          ("poseidonB_trained", "bf16x3"), ("poseidonT_trained", "fp16"), ("poseidonT_hf", "fp16"),
                                          ("poseidonB_trained", "fp16"), ("poseidonB_hf", "fp16"),
                                          # BASELINE config 5: Poseidon-B at 256x256 (shifted 16x16 windows at stages 0 AND 1)
                                          ("poseidonB256_trained", "fp32"), ("poseidonB256_trained", "fp16"),
                                          ("poseidonB256_trained", "bf16")])
def test_poseidon_presets(name, compute):
    f, meta = load_fixture(name)
    cfg, model = build(meta, compute)
    out = model(**inputs(cfg, meta))
    out.loss.backward()
    torch.cuda.synchronize()
    e_out = rel_l2(out.output.detach().cpu().numpy(), f["output"])
    e_loss = abs(float(out.loss.detach()) - float(f["loss"])) / abs(float(f["loss"]))
    names = [str(n) for n in f["grad_names"]]
    ref_norm = f["grad_norms"]
    mine = {k: float(p.grad.double().norm()) for k, p in model.named_parameters()}
    dev = np.array([abs(mine[n] - r) / max(r, 1e-12) for n, r in zip(names, ref_norm) if r > 1e-7])
    print(f"\n[{name} {compute}] out rel-L2 {e_out:.2e} loss rel {e_loss:.2e} grad-norm dev median {np.median(dev):.2e} max {dev.max():.2e}")
    if compute == "fp32":
        assert e_out < 1e-5 + 5e-6
        assert e_loss < 2e-5
        assert np.median(dev) < 1e-4
        grads_report(model, f, tol_each=1e-3, tol_global=1e-3)
    elif compute == "fp16":
        # THE north-star bound for 16-bit operands: 1e-3 output rel-L2 on both parameter regimes (binary16 operands, fp32
        # accumulation / statistics / residual stream; predicted 7.1e-4..7.9e-4 trained-like, 1.7e-4 HF-init by
        # tools/probes/precision_sim.py).  Gradients: the backward runs under a power-of-two gradient scale, no overflow.
        assert e_out < 1e-3 and e_loss < 1e-3
        assert int(model._engine.grad_overflow) == 0
        assert np.median(dev) < 1.5e-3         # measured 3.8e-4 .. 6.0e-4
        # stored full gradients, global rel-L2: measured 1.1e-3 (T) / 2.6e-3 (B) / 1.6e-3 (B@256²) trained-like, 3e-4 / 1e-4 HF-init
        # per tensor (round 3, MI355X): worst 2.9e-2 (T trained: a key.weight) / 2.8e-3 (B trained) / 5.1e-3 (B@256²) / 4.3e-2 (B HF-init: the last
        # layer of a bias MLP); tensors whose true gradient is round-off in the reference itself (|g| < 1e-6 absolute) fall under `floor`
        # (round 5, re-measured: global 1.2e-3 / 1.8e-3 / 1.9e-3 trained-like, 3.2e-4 / 1.1e-4 HF-init; worst tensor 4.0e-2 / 3.4e-3 / 6.1e-3 / 4.9e-2)
        g, worst = grads_report(model, f, tol_each=8e-2, tol_global=4e-3, floor=1e-6)       # (logit_scale included since round 6: measured 3e-5 .. 5.6e-3, tools/probes/logit_scale_grad.py)
        print(f"[{name} fp16] stored gradients: global rel-L2 {g:.2e}, worst {worst}")
        # the ConvNeXt skip blocks' branch gradients (behind the layer scale: 1e-6 in the HF-init regime, ~2^-20 below the rest)
        # survive binary16 through the device-side local power-of-two rescale (engine.convnext_bwd)
        for k, p in model.named_parameters():
            if k.startswith("residual_blocks.") and "grad:" + k in f.files and float(np.linalg.norm(f["grad:" + k])) > 0:
                assert rel_l2(p.grad.detach().cpu().numpy(), f["grad:" + k]) < 5e-2, k
    elif compute == "bf16x3":
        # fp32 operands split into hi + lo bf16 (three bf16 MFMAs per product): the north star's 1e-3 bound for the bf16 path,
        # with margin — on BOTH parameter regimes
        assert e_out < 1e-4 and e_loss < 1e-4
        assert np.median(dev) < 1e-3
        grads_report(model, f, tol_each=2e-2, tol_global=2e-3)
    else:
        assert e_out < (3e-3 if meta["regime"] == "hf" else 2e-2)  # measured 1.4e-3..2e-3 / 6.3e-3..6.7e-3, DESIGN.md "Numerics"


@pytest.mark.parametrize("compute", ["fp32", "fp16"])
def test_poseidon_L_config4(compute):
    """BASELINE.json config 4's shape: Poseidon-L (embed_dim 192: head_dim 64, C up to 1536, 629 M parameters), 5→5 channels with
    loss groups [0,1,3,4,5].  Parameters are generated on the GPU (bit-identical to the host generator, asserted on one tensor)
    and the modules are constructed there, which keeps this test at a few seconds."""
    f, meta = load_fixture("poseidonL_trained")
    cfg = ScOTConfig(**meta["cfg"])
    shapes = param_shapes(cfg)
    with torch.device(DEV):
        model = ScOT(cfg, compute=compute, engine_options={"fused_min_rows": 0})      # (batch 1: stay on the fused tail of stage 0, as `build`)
    with generate_on(DEV):
        sd = synth_state_dict(shapes, meta["regime"])
    for k in ("encoder.layers.2.blocks.3.intermediate.dense.weight", "embeddings.norm.weight.weight"):
        assert torch.equal(sd[k].cpu(), synth_state_dict({k: shapes[k]}, meta["regime"])[k]), k
    model.load_state_dict(sd)
    del sd
    out = model(**inputs(cfg, meta))
    out.loss.backward()
    torch.cuda.synchronize()
    e_out = rel_l2(out.output.detach().cpu().numpy(), f["output"])
    e_loss = abs(float(out.loss.detach()) - float(f["loss"])) / abs(float(f["loss"]))
    names = [str(n) for n in f["grad_names"]]
    mine = {k: float(p.grad.double().norm()) for k, p in model.named_parameters()}
    dev = np.array([abs(mine[n] - r) / max(r, 1e-12) for n, r in zip(names, f["grad_norms"]) if r > 1e-7])
    print(f"\n[poseidonL {compute}] out rel-L2 {e_out:.2e} loss rel {e_loss:.2e} grad-norm dev median {np.median(dev):.2e} max {dev.max():.2e}")
    if compute == "fp16":       # the headline mode on config 4's model (head_dim 64): the north star's 1e-3
        assert e_out < 1e-3 and e_loss < 1e-3 and np.median(dev) < 1.5e-3 and int(model._engine.grad_overflow) == 0      # (measured 6.2e-4 / 8.9e-6 / 4.1e-4)
        return
    assert e_out < 1e-5 + 5e-6
    assert e_loss < 2e-5
    assert np.median(dev) < 1e-4
    grads_report(model, f, tol_each=1e-3, tol_global=1e-3)


@pytest.mark.parametrize("size", [64, 16])
def test_spectral_resize_path(size):
    f, meta = load_fixture(f"tiny_resize{size}")
    cfg, model = build(meta, "fp32")
    with torch.no_grad():
        out = model(**inputs(cfg, meta))
    assert rel_l2(out.output.cpu().numpy(), f["output"]) < 2e-5
    assert abs(float(out.loss) - float(f["loss"])) < 5e-5 * abs(float(f["loss"]))


def test_vs_cpu_oracle_other_config():
    """A configuration with no reference fixture: 5→3 channels, window 8, heads [2,4], p=2 grouped loss."""
    from oracle import scot_cpu
    kw = dict(image_size=64, patch_size=4, num_channels=5, num_out_channels=3, embed_dim=32, depths=[2, 2], num_heads=[2, 4],
              skip_connections=[1, 1], window_size=8, mlp_ratio=2.0, qkv_bias=True, drop_path_rate=0.0, p=2,
              channel_slice_list_normalized_loss=[0, 2, 3], use_conditioning=True, learn_residual=True)
    meta = dict(cfg=kw, regime="trained", batch=3, kind="smooth")
    cfg, model = build(meta, "fp32")
    kwargs = inputs(cfg, meta)
    out = model(**kwargs)
    out.loss.backward()
    sd = {k: v.double().requires_grad_(True) for k, v in synth_state_dict(param_shapes(cfg), "trained").items()}
    loss, pred = scot_cpu.scot_forward(sd, cfg, kwargs["pixel_values"].cpu().double(), kwargs["time"].cpu().double(),
                                       kwargs["labels"].cpu().double())
    loss.backward()
    assert rel_l2(out.output.detach().cpu().numpy(), pred.detach().numpy()) < 1e-5
    assert abs(float(out.loss.detach()) - float(loss.detach())) < 1e-5 * abs(float(loss.detach()))
    num = den = 0.0
    for k, p in model.named_parameters():
        ref = sd[k].grad.numpy()
        err = np.linalg.norm(p.grad.cpu().numpy().astype(np.float64) - ref)
        assert err < 5e-4 * np.linalg.norm(ref) + 1e-9, k
        num += err ** 2
        den += np.linalg.norm(ref) ** 2
    assert (num / den) ** 0.5 < 5e-5


def test_grad_accumulation_and_eval_mode():
    f, meta = load_fixture("tiny_trained")
    cfg, model = build(meta, "fp32")
    kw = inputs(cfg, meta)
    model(**kw).loss.backward()
    g1 = model.flat_grads().clone()
    model(**kw).loss.backward()          # accumulates (+=) like autograd
    assert rel_l2(model.flat_grads().cpu().numpy(), (2 * g1).cpu().numpy()) < 1e-5
    model.zero_grad()
    assert float(model.flat_grads().abs().sum()) == 0.0
    with torch.no_grad():
        o = model(**kw)
    assert rel_l2(o.output.cpu().numpy(), f["output"]) < 1.5e-5
    assert o.loss.grad_fn is None


def test_ar_rollout_matches_reference_trainer():
    """reference Trainer._model_forward (trainer.py:452-603) pins generated by the real reference (rollout_tiny.npz)."""
    import os
    from conftest import GOLDEN
    from poseidon_amd.harness import rollout
    f = np.load(os.path.join(GOLDEN, "rollout_tiny.npz"))
    fx, meta = load_fixture("tiny_trained")
    cfg, model = build(meta, "fp32")
    model.eval()
    pv, t, lab = synth_inputs(2, 4, 4, 32, "smooth")
    kw = dict(pixel_values=pv.to(DEV), time=t.to(DEV), labels=lab.to(DEV))
    with torch.no_grad():
        o = rollout(model, kw, 3)
        # 3 network applications chained: the per-call 1e-6 deviation is amplified by the (trained-like) network's gain
        assert rel_l2(o.output.cpu().numpy(), f["int3_output"]) < 1e-3
        assert abs(float(o.loss) - float(f["int3_loss"])) < 1e-3 * abs(float(f["int3_loss"]))
        o = rollout(model, kw, 2, output_all_steps=True)
        assert tuple(o.output.shape) == (2, 2, 4, 32, 32) and tuple(o.loss.shape) == (2,)
        assert rel_l2(o.output.cpu().numpy(), f["int2all_output"]) < 1e-3
        assert rel_l2(o.loss.cpu().numpy(), f["int2all_loss"]) < 1e-3
        o = rollout(model, dict(kw, time=kw["time"] * 0.5), [1, 2])
        assert rel_l2(o.output.cpu().numpy(), f["list12_output"]) < 1e-3
        assert abs(float(o.loss) - float(f["list12_loss"])) < 1e-3 * abs(float(f["list12_loss"]))


@pytest.mark.parametrize("name", ["tiny_trained", "tiny_learnres_mask", "tiny_w16"])
def test_step_tape_replay_matches_direct_launches(name):
    """Step tape (engine.forward/backward): step 1 runs the ops, step 2 records, steps 3+ replay the recorded launches on
    NEW inputs.  Every step must give the loss / prediction / gradients of a model that never tapes."""
    f, meta = load_fixture(name)
    cfg, taped = build(meta, "fp32")
    _, plain = build(meta, "fp32")
    kw = inputs(cfg, meta)
    for step in range(5):
        kws = {k: (v if v.dtype == torch.bool else v * (1.0 + 0.25 * step) + 0.01 * step) for k, v in kw.items()}
        outs = []
        for m in (taped, plain):
            if m._engine is not None:
                m._engine.tape_mode = m is taped
            m.zero_grad()
            out = m(**kws)
            (out.loss * (1.0 + step)).backward()     # a different upstream gradient every step
            outs.append((float(out.loss), out.output.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}))
        if step == 0:
            continue   # engines exist from here on (step 0 of `plain` may have run with the default tape_mode: it only warms)
        (l0, o0, g0), (l1, o1, g1) = outs
        assert abs(l0 - l1) <= 1e-6 * abs(l1), (step, l0, l1)
        assert rel_l2(o0.cpu().numpy(), o1.cpu().numpy()) < 1e-6, step
        for k in g0:
            n = float(g1[k].norm())
            assert float((g0[k] - g1[k]).norm()) <= 2e-5 * n + 1e-9, (step, k)
    ent = [e for e in taped._engine._taped.values() if e["state"] == "ready"]
    assert len(ent) == 1 and len(ent[0]["fwd"]) > 20 and len(ent[0]["bwd"][False][0]) > 20      # (one recorded backward per gradient-commit variant; eager zero_grad: the `add` one)
    assert not plain._engine._taped or all(e["state"] == "warm" for e in plain._engine._taped.values())


def test_stochastic_depth_matches_reference_draws():
    """drop_path_rate > 0 in training: the per-sample scale goes through the LN kernels; with the
    reference's recorded keep masks injected the step must reproduce the reference's training-mode step."""
    f, meta = load_fixture("tiny_droppath")
    cfg, model = build(meta, "fp32")
    kw = inputs(cfg, meta)
    model.train()
    model(**kw).loss.backward()          # builds the engine (random draws: only checks that it runs and differs from eval)
    eng = model._engine
    eng.tape_mode = False
    eng.drop_path_masks = {}
    for k in f.files:
        if k.startswith("mask:"):
            _, name, which = k.split(":")
            eng.drop_path_masks[(name, int(which))] = torch.from_numpy(f[k])
    model.zero_grad()
    out = model(**kw)
    out.loss.backward()
    assert rel_l2(out.output.detach().cpu().numpy(), f["output"]) < 1e-5
    assert abs(float(out.loss) - float(f["loss"])) < 1e-5 * abs(float(f["loss"]))
    grads_report(model, f, tol_each=5e-3, tol_global=1e-4)   # global 5.5e-5 (the CPU oracle sits at 6.9e-5 from this fixture); worst tensor: a logit_scale (cancelling sum, |g| = 2.6e-3): 2.5e-3
    # random draws: training differs from eval, eval is deterministic and ignores the rate
    eng.drop_path_masks = None
    model.eval()
    with torch.no_grad():
        e1, e2 = model(**kw).output.clone(), model(**kw).output.clone()
    assert torch.equal(e1, e2)
    model.train()
    diffs = []
    for _ in range(4):
        with torch.enable_grad():
            o = model(**kw)
        o.loss.backward()
        diffs.append(float((o.output.detach() - e1).abs().max()))
    assert max(diffs) > 1e-3            # 14 branches x 4 samples at rates up to 0.5: some branch is dropped


def test_fused_adamw_matches_torch_adamw_with_clipping():
    """scOT.trainer.FusedAdamW (3 launches over the arenas) vs torch.optim.AdamW on the reference's parameter groups +
    torch.nn.utils.clip_grad_norm_ (HF Trainer's step): same parameters after 4 steps."""
    from scOT.trainer import FusedAdamW, create_optimizer
    f, meta = load_fixture("tiny_trained")
    cfg, ma = build(meta, "fp32")
    _, mb = build(meta, "fp32")
    kw = inputs(cfg, meta)
    gk = dict(learning_rate_embedding_recovery=3e-3, learning_rate_time_embedding=2e-3)
    ma(**kw).loss.backward()      # engines / arenas exist
    mb(**kw).loss.backward()
    fa = FusedAdamW(ma, lr=5e-3, weight_decay=0.05, max_grad_norm=0.05, **gk)
    tb = create_optimizer(mb, 5e-3, weight_decay=0.05, **gk)
    assert [len(g["params"]) for g in fa.param_groups] == [len(g["params"]) for g in tb.param_groups]
    for step in range(4):
        fa.zero_grad()
        (ma(**kw).loss * (1.0 + step)).backward()
        # identical gradients for both optimizers: Adam's first steps are ~lr·sign(g), so the 1e-7 run-to-run noise of the
        # atomically accumulated gradients would flip near-zero entries and mask what is compared here (the step arithmetic)
        mb._prepare_grads()
        mb._arena.grad.copy_(ma._arena.grad)
        norm_b = torch.nn.utils.clip_grad_norm_(mb.parameters(), 0.05)
        fa.param_groups[0]["lr"] = tb.param_groups[0]["lr"] = 5e-3 * (1.0 - 0.1 * step)    # what a scheduler does
        fa.step()
        tb.step()
        assert abs(float(fa.last_grad_norm) - float(norm_b)) < 1e-5 * float(norm_b)
        assert float(norm_b) > 0.05                                                          # clipping is active
    worst = 0.0
    for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        d = float((pa - pb).abs().max()) / (float(pb.abs().max()) + 1e-12)
        worst = max(worst, d)
        assert d < 2e-5, (n, d)
    # the zero slot of the fused qkv bias (bias-free key projection) is untouched
    a = ma._arena
    pre = "encoder.layers.0.blocks.0.attention.self."
    o, c = a.offsets[pre + "qkv_bias"], ma.config.embed_dim
    assert float(a.data[o + c:o + 2 * c].abs().max()) == 0.0


def test_training_rollout_with_grad_matches_oracle():
    """VERDICT r1 weak #4 / ADVICE (high): the reference's AR *training* step runs n forwards, sums the losses and calls ONE
    backward (trainer.py:466-490, 605-635) — forward-forward-backward through the step tape.  Gradients must equal the
    oracle's doing the same, on the warm, the recording and the replaying step alike."""
    from oracle import scot_cpu
    from poseidon_amd.harness import rollout
    f, meta = load_fixture("tiny_trained")
    cfg, model = build(meta, "fp32")
    model.train()
    pv, t, lab = synth_inputs(2, 4, 4, 32, "smooth")
    sd = {k: v.clone().requires_grad_(True) for k, v in synth_state_dict(param_shapes(cfg), meta["regime"]).items()}
    l1, o1 = scot_cpu.scot_forward(sd, cfg, pv, t / 2, lab)
    l2, o2 = scot_cpu.scot_forward(sd, cfg, o1.detach(), t / 2, lab)
    ((l1 + l2) / 2).backward()
    kw = dict(pixel_values=pv.to(DEV), time=t.to(DEV), labels=lab.to(DEV))
    for step in range(4):     # warm, record, replay, replay
        model.zero_grad()
        out = rollout(model, kw, 2)
        out.loss.backward()
        torch.cuda.synchronize()
        # two chained applications: the per-call 1e-6 deviation is amplified by the (trained-like) network's gain, as in
        # test_ar_rollout_matches_reference_trainer
        assert abs(float(out.loss) - float((l1 + l2) / 2)) < 1e-4 * abs(float(l1 + l2) / 2), step
        assert rel_l2(out.output.detach().cpu().numpy(), o2.detach().numpy()) < 1e-3, step
        num = den = 0.0
        for k, p in model.named_parameters():
            num += float((p.grad.cpu().double() - sd[k].grad.double()).norm()) ** 2
            den += float(sd[k].grad.double().norm()) ** 2
        assert (num / den) ** 0.5 < 1e-3, (step, (num / den) ** 0.5)   # (the aliasing bug this guards against gave 0.5)


def test_outputs_are_fresh_tensors_across_steps():
    """`out.output` / `out.loss` of step N must still hold step N's values after step N+1 (no aliasing of recorded buffers)."""
    f, meta = load_fixture("tiny_trained")
    cfg, model = build(meta, "fp32")
    kw = inputs(cfg, meta)
    kept = []
    for step in range(5):
        kws = {k: (v if v.dtype == torch.bool else v * (1.0 + 0.3 * step)) for k, v in kw.items()}
        model.zero_grad()
        out = model(**kws)
        out.loss.backward()
        kept.append((out.output.detach(), out.output.detach().clone(), out.loss.detach(), float(out.loss)))
    torch.cuda.synchronize()
    for held, copy, lheld, lval in kept:
        assert torch.equal(held, copy) and float(lheld) == lval
    assert not torch.equal(kept[3][0], kept[4][0])


def test_eval_mode_with_grad_enabled_is_deterministic():
    """stochastic depth follows module.training, not torch.is_grad_enabled (HF:565-586).  eval() with
    gradients enabled and drop_path_rate > 0 gives the deterministic output AND gradients; train() under no_grad still draws."""
    f, meta = load_fixture("tiny_droppath")
    cfg, model = build(meta, "fp32")
    assert cfg.drop_path_rate > 0
    kw = inputs(cfg, meta)
    model.eval()
    with torch.no_grad():
        ref = model(**kw).output.clone()
    outs = []
    for _ in range(3):
        model.zero_grad()
        o = model(**kw)                       # grad enabled
        o.loss.backward()
        outs.append(o.output.detach().clone())
    assert all(torch.equal(x, outs[0]) for x in outs) and rel_l2(outs[0].cpu().numpy(), ref.cpu().numpy()) < 1e-6
    assert float(model.flat_grads().abs().sum()) > 0
    model.train()
    with torch.no_grad():
        draws = [model(**kw).output.clone() for _ in range(4)]
    assert max(float((d - ref).abs().max()) for d in draws) > 1e-3


def test_tuple_return_layout_matches_reference():
    """return_dict=False: (loss,) + (prediction,) + decoder_output[1:] + encoder_outputs[1:] (reference model.py:1486-1488): the
    encoder's hidden states are always the last element, the decoder's only with output_hidden_states=True."""
    f, meta = load_fixture("tiny_trained")
    cfg, model = build(meta, "fp32")
    kw = inputs(cfg, meta)
    with torch.no_grad():
        d = model(**kw, output_hidden_states=True)
        t0 = model(**kw, return_dict=False)
        t1 = model(**kw, return_dict=False, output_hidden_states=True)
    nl = len(cfg.depths)
    assert len(t0) == 3 and len(t1) == 4
    # (the loss sums are accumulated with atomics: the last bit depends on the order the workgroups arrive in)
    assert abs(float(t0[0]) - float(d.loss)) <= 1e-6 * abs(float(d.loss)) and torch.equal(t0[1], d.output)
    assert isinstance(t0[2], tuple) and len(t0[2]) == nl + 1 and len(t1[2]) == nl + 1 and len(t1[3]) == nl + 1
    assert len(d.hidden_states) == 2 * (nl + 1)
    for a, b in zip(t1[2] + t1[3], d.hidden_states):
        assert torch.equal(a, b)
    for a, b in zip(t0[2], d.hidden_states[nl + 1:]):
        assert torch.equal(a, b)


def test_nonzero_dropout_is_refused():
    f, meta = load_fixture("tiny_trained")
    with pytest.raises(NotImplementedError):
        ScOT(ScOTConfig(**dict(meta["cfg"], hidden_dropout_prob=0.1)))


def test_metrics_on_device_tensors_match_reference_pins():
    """the evaluation metrics run on the GPU tensors a rollout returns (only the error vectors cross PCIe);
    same pins as the CPU test (values produced by the reference's metrics module, tests/golden/make_metrics_pins.py)."""
    import json
    import os
    from conftest import GOLDEN
    from poseidon_amd.synth import closed_form_tensor
    from scOT import metrics as M
    pins = json.load(open(os.path.join(GOLDEN, "metrics_pins.json")))
    pr = torch.as_tensor(np.asarray(closed_form_tensor("metrics:pred", (6, 4, 16, 16), 1.0), dtype=np.float32)).to(DEV)
    tg = np.asarray(closed_form_tensor("metrics:target", (6, 4, 16, 16), 1.0), dtype=np.float32)
    tg[3] = 0.0
    tg = torch.as_tensor(tg).to(DEV)

    def close(a, b, tol=3e-6):
        a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        return np.all(np.abs(a - b) <= tol * np.maximum(np.abs(b), 1e-30))
    for p in (1, 2):
        e = M.lp_error(pr, tg, p=p)
        assert isinstance(e, torch.Tensor) and e.is_cuda and close(e, pins[f"lp_error_p{p}"])
        assert close(M.relative_lp_error(pr, tg, p=p), pins[f"relative_lp_error_p{p}"])
        assert close(M.mean_relative_lp_error(pr, tg, p=p), pins[f"mean_relative_p{p}"])
        assert close(M.median_relative_lp_error(pr, tg, p=p), pins[f"median_relative_p{p}"])
    out = M.channel_group_metrics(pr, tg, [0, 1, 3, 4], ["rho", "uv", "p"])
    for i, n in enumerate(["rho", "uv", "p"]):
        g = pins[f"group{i}"]
        assert close(out[n + "/median_relative_l1_error"], g["median_rel"]) and close(out[n + "/mean_relative_l1_error"], g["mean_rel"])
        assert close(out[n + "/max_relative_l1_error"], g["max_rel"]) and close(out[n + "/median_l1_error"], g["median_abs"])


def test_mask_tokens_match_reference():
    """ScOT(config, use_mask_token=True) + bool_masked_pos (reference model.py:323-327, 353-359) against the real
    reference's fixture, through the module API (output, loss, every gradient incl. the mask token's)."""
    from poseidon_amd.synth import synth_token_mask
    f, meta = load_fixture("tiny_masktoken")
    cfg = ScOTConfig(**meta["cfg"])
    model = ScOT(cfg, use_mask_token=True, compute="fp32")
    model.load_state_dict(synth_state_dict(param_shapes(cfg, use_mask_token=True), meta["regime"]))
    model = model.to(DEV)
    kw = inputs(cfg, meta)
    bmp = synth_token_mask(meta["batch"], (cfg.image_size // cfg.patch_size) ** 2).to(DEV)
    out = model(**kw, bool_masked_pos=bmp)
    out.loss.backward()
    torch.cuda.synchronize()
    assert rel_l2(out.output.detach().cpu().numpy(), f["output"]) < 1e-5 + 5e-6
    assert abs(float(out.loss.detach()) - float(f["loss"])) < 2e-5 * abs(float(f["loss"]))
    grads_report(model, f, tol_each=1e-3, tol_global=1e-4)
    with pytest.raises(ValueError):
        build(meta, "fp32")[1](**kw, bool_masked_pos=bmp)      # a model without the mask token refuses masked positions


def test_device_resident_dataset_batches():
    """trajectories resident in HBM, a batch = one scot_gather_pairs launch; equals the reference-style
    __getitem__ samples (CPU path) and feeds the model directly."""
    from scOT.problems.base import get_dataset
    rng = np.random.default_rng(0)
    rd = {"data": rng.standard_normal((12, 21, 5, 32, 32)).astype(np.float32)}
    ds = get_dataset("fluids.compressible.Riemann", which="train", num_trajectories=5, reader=rd, n_max=12, n_val=4, n_test=3)
    ds.resolution = 32
    dev = ds.to_device(DEV)
    idx = [0, 7, 35, 36, 100, 179]
    b = dev.batch(idx)
    for k, j in enumerate(idx):
        s = ds[j]
        assert np.allclose(b["pixel_values"][k].cpu().numpy(), s["pixel_values"].numpy(), rtol=1e-6, atol=1e-6)
        assert np.allclose(b["labels"][k].cpu().numpy(), s["labels"].numpy(), rtol=1e-6, atol=1e-6)
        assert float(b["time"][k]) == pytest.approx(s["time"])
    f, meta = load_fixture("tiny_trained")
    cfg, model = build(meta, "fp32")              # 32x32, 4 -> 4 channels: the batch goes straight into the model
    out = model(pixel_values=b["pixel_values"], time=b["time"], labels=b["labels"], pixel_mask=b["pixel_mask"])
    assert bool(torch.isfinite(out.loss))


@pytest.mark.parametrize("name", ["wave.Layer", "elliptic.Helmholtz.time", "fluids.compressible.steady.Airfoil",
                                  "fluids.incompressible.forcing.KolmogorovFlow"])
def test_device_resident_batches_other_families(name):
    """The non-fluids readers on the GPU: scot_gather_planes (static, scalar and analytic source planes; inputs and labels with
    different recipes) == the reference-style __getitem__ samples; the incompressible readers' `resolution=` through the native
    spectral resize."""
    from scOT.problems.base import get_dataset
    rng = np.random.default_rng(1)
    R = 128 if "Kolmogorov" in name else 32
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    rd = {"wave.Layer": lambda: {"solution": f(12, 21, R, R), "c": f(12, R, R)},
          "elliptic.Helmholtz.time": lambda: {"a": f(12, R, R), "bc": f(12), "u": f(12, R, R)},
          "fluids.compressible.steady.Airfoil": lambda: {"solution": np.concatenate([(rng.random((12, 1, R, R)) > 0.6).astype(np.float32), f(12, 1, R, R)], 1)},
          "fluids.incompressible.forcing.KolmogorovFlow": lambda: {"solution": f(12, 21, 2, R, R)}}[name]()
    ds = get_dataset(name, which="train", num_trajectories=5, reader=rd, n_max=12, n_val=4, n_test=3)
    ds.resolution = R
    dev = ds.to_device(DEV)
    idx = [0, 4, 2] if ds.steady else [0, 7, 35, 36, 100, 179]
    b = dev.batch(idx)
    for k, j in enumerate(idx):
        s = ds[j]
        assert set(b) == set(s)
        assert np.allclose(b["pixel_values"][k].cpu().numpy(), s["pixel_values"].numpy(), rtol=1e-6, atol=1e-6)
        assert np.allclose(b["labels"][k].cpu().numpy(), s["labels"].numpy(), rtol=1e-6, atol=1e-6)
        if "time" in s:
            assert float(b["time"][k]) == pytest.approx(s["time"])
        if "pixel_mask" in s:
            assert torch.equal(b["pixel_mask"][k].cpu(), s["pixel_mask"])


def _recipe_pins_for_device():
    pins = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_recipe_pins.json")))
    # (the "val" pins take the whole validation split: 120 synthetic trajectories to generate per dataset — train / test pins cover the recipes)
    return [p for p in pins if p["which"] in ("train", "test") and "raises" not in p]


@pytest.mark.parametrize("pin", _recipe_pins_for_device(), ids=lambda p: f"{p['name']}-{p['which']}" + ("-" + "_".join(p["kw"]) if p["kw"] else ""))
def test_device_batches_match_the_reference_readers(pin):
    """HBM-resident batches (scot_gather_pairs / scot_gather_planes, native spectral resize) against the pins the reference's own
    readers produced on the same synthetic files (tests/golden/make_dataset_recipe_pins.py)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_data_cpu as T
    tol = 2e-6 if "resolution" not in pin["kw"] else 5e-5
    T.check_against_pin(pin, lambda ds, i: T.device_rows(ds, i, DEV), tol=tol)


def test_device_resident_batches_downsampled():
    from scOT.problems.base import get_dataset
    rng = np.random.default_rng(2)
    rd = {"velocity": rng.standard_normal((12, 21, 2, 128, 128)).astype(np.float32)}
    ds = get_dataset("fluids.incompressible.Sines", which="train", num_trajectories=5, reader=rd, n_max=12, n_val=4, n_test=3, resolution=64)
    b = ds.to_device(DEV).batch([3, 40])
    for k, j in enumerate([3, 40]):
        s = ds[j]
        assert b["pixel_values"].shape[-2:] == (64, 64)
        assert np.allclose(b["pixel_values"][k].cpu().numpy(), s["pixel_values"].numpy(), rtol=1e-4, atol=2e-5)
        assert np.allclose(b["labels"][k].cpu().numpy(), s["labels"].numpy(), rtol=1e-4, atol=2e-5)


def test_fused_adamw_skips_overflowed_steps_and_adapts_the_gradient_scale():
    """fp16 compute mode: a step whose (reduced) gradient norm is not finite leaves parameters and moments untouched, does not
    advance Adam's clock, and halves the engine's gradient scale; N clean steps in a row double it (torch.cuda.amp.GradScaler's
    rule) — all ON THE DEVICE (scot_clip_coef / scot_adamw_step / scot_optim_finish), no host round trip in step()."""
    from scOT.trainer import FusedAdamW
    f, meta = load_fixture("tiny_trained")
    cfg, model = build(meta, "fp16")
    kw = inputs(cfg, meta)
    model(**kw).loss.backward()
    opt = FusedAdamW(model, lr=1e-2)
    opt.growth_interval = 2
    eng = model._engine
    S0 = eng.grad_scale_value()
    assert S0 == 2.0 ** int(np.floor(np.log2(kw["labels"].numel()))) and opt.loss_scale_value() == S0
    p0 = model.flat_parameters().clone()
    opt.step()                                            # a normal step moves the parameters
    torch.cuda.synchronize()
    p1 = model.flat_parameters().clone()
    assert not torch.equal(p0, p1) and int(eng.grad_overflow) == 0 and (opt.applied_steps(), opt.skipped_steps()) == (1, 0)
    opt.zero_grad()
    model(**kw).loss.backward()
    model.flat_grads()[model._arena.offsets["embeddings.norm.bias.bias"]] = float("inf")     # as an overflow in the 16-bit backward leaves it
    m1 = opt.exp_avg.clone()
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(model.flat_parameters(), p1) and torch.equal(opt.exp_avg, m1)      # skipped
    assert (opt.applied_steps(), opt.skipped_steps()) == (1, 1) and eng.grad_scale_value() == S0 / 2
    for k in range(2):                                    # two clean steps in a row: the scale is doubled again
        opt.zero_grad()
        out = model(**kw)
        out.loss.backward()
        opt.step()
    torch.cuda.synchronize()
    assert not torch.equal(model.flat_parameters(), p1) and (opt.applied_steps(), opt.skipped_steps()) == (3, 1)
    assert eng.grad_scale_value() == S0 and int(eng.grad_overflow) == 0
    # the gradients of a step are the same at any (power-of-two) scale: halving the scale by hand changes nothing but round-off
    opt.zero_grad()
    model(**kw).loss.backward()
    g_full = model.flat_grads().clone()
    eng.scale_state.copy_(torch.tensor([S0 / 4, 4 / S0, 0.0, 0.0]))
    opt.zero_grad()
    model(**kw).loss.backward()
    assert float((model.flat_grads() - g_full).norm() / g_full.norm()) < 2e-3
    # a checkpoint carries the clock and the scale
    sd = opt.state_dict()
    assert sd["fused"]["step_state"].tolist() == [3, 1] and float(sd["fused"]["scale_state"][0]) == S0 / 4


def test_weight_copies_follow_the_master_weights():
    """The 16-bit operand copies of the weights are refreshed only when the fp32 master changed: (a) in-place edits through torch
    (version counters) and `mark_weights_dirty()` are seen by the next forward, (b) the fused AdamW writes the copies itself —
    the forward after it issues no cast and still computes with the new weights."""
    from scOT.trainer import FusedAdamW
    f, meta = load_fixture("tiny_trained")
    cfg, model = build(meta, "fp16")
    kw = inputs(cfg, meta)
    eng_casts = []
    real_cast = ops.cast

    def counting_cast(src, dst):
        if src.numel() == model._arena.size:        # the weight arena (activations are cast through the same entry point)
            eng_casts.append(src.numel())
        return real_cast(src, dst)
    ops.cast = counting_cast
    try:
        with torch.no_grad():
            y0 = model(**kw).output.clone()
            n0 = len(eng_casts)
            y1 = model(**kw).output.clone()
            assert len(eng_casts) == n0 and torch.equal(y0, y1)                  # unchanged weights: no cast
            name = "patch_recovery.projection.bias"
            dict(model.named_parameters())[name].add_(0.5)                        # in-place edit through torch
            y2 = model(**kw).output.clone()
            assert len(eng_casts) > n0 and float((y2 - y1).abs().mean()) > 0.1
            w = dict(model.named_parameters())["encoder.layers.0.blocks.0.output.dense.weight"]
            w.data.mul_(1.5)                                                       # behind torch's back ...
            n1 = len(eng_casts)
            model.mark_weights_dirty()                                             # ... so the engine has to be told
            y3 = model(**kw).output.clone()
            assert len(eng_casts) > n1 and not torch.equal(y3, y2)
        model(**kw).loss.backward()
        opt = FusedAdamW(model, lr=1e-2)
        n2 = len(eng_casts)
        opt.step()
        eng = model._engine
        torch.cuda.synchronize()
        # the optimizer's copy == a fresh cast of the new master weights, for every parameter; transposed copies too
        fresh = torch.empty_like(eng.shadow)
        prev_kind = ops.use(eng.lib_kind)
        real_cast(eng.arena.data, fresh)
        ops.use(prev_kind)
        for nme in ("encoder.layers.0.blocks.0.output.dense.weight", "embeddings.norm.weight.bias", "decoder.layers.1.blocks.1.attention.self.value.weight"):
            o, k = eng.arena.offsets[nme], eng.arena.numel(nme)
            assert torch.equal(eng.shadow[o:o + k], fresh[o:o + k])
        wn = next(n for n in eng._wt_names if not n.endswith("qkv_weight"))        # (matrices of at least 32 x 32 keep a transposed copy)
        assert torch.equal(eng.WT(wn), eng.W(wn).t().contiguous())
        with torch.no_grad():
            y4 = model(**kw).output.clone()
        assert len(eng_casts) == n2 and not torch.equal(y4, y3)                   # no cast after the fused step, new weights in use
        model2_sd = {k: v.clone() for k, v in model.state_dict().items()}
    finally:
        ops.cast = real_cast
    cfg2, fresh_model = build(meta, "fp16")
    fresh_model.load_state_dict(model2_sd)
    with torch.no_grad():
        assert torch.equal(fresh_model(**kw).output, y4)                          # == a model that casts everything from scratch


@pytest.mark.parametrize("regime_fixture,lr,tol", [("tiny_trained", 1e-4, 1e-2), ("tiny_hf", 2e-3, 1e-4)])
def test_short_training_run_fp16_tracks_fp32(regime_fixture, lr, tol):
    """End to end through everything the fp16 mode adds (global gradient scale, per-range un-scale, device-side rescale of the
    ConvNeXt branches, overflow-skipping fused AdamW, the step tape): 8 optimiser steps from the same initial state in fp32 and
    in fp16 on the same batch — the loss falls and the two trajectories stay together (the reference trains in fp32).  Measured
    (tools/probes/train_traj_probe.py, 4 repeats): trained-like tiny model at lr 1e-4: max relative gap 1.1e-3..3.9e-3 (the
    1e-5-accurate bf16x3 mode: 1.2e-3; fp32 against ITSELF 4e-7).  At lr 3e-4 that model is chaotic under AdamW — two fp32 runs
    differ by up to 1.6e-3 through the atomics' summation order alone, fp16 by 1e-2..2e-2 — so the test stays below it.  HF-init: 3e-6."""
    from scOT.trainer import FusedAdamW
    f, meta = load_fixture(regime_fixture)
    traj = {}
    for compute in ("fp32", "fp16"):
        cfg, model = build(meta, compute)
        kw = inputs(cfg, meta)
        opt = FusedAdamW(model, lr=lr, weight_decay=0.01, max_grad_norm=5.0)      # (before any forward: creates the arenas itself)
        losses = []
        for _ in range(8):
            opt.zero_grad()
            out = model(**kw)
            out.loss.backward()
            opt.step()
            losses.append(float(out.loss.detach()))
        torch.cuda.synchronize()
        if compute == "fp16":
            assert int(model._engine.grad_overflow) == 0
        traj[compute] = np.array(losses)
    a, b = traj["fp32"], traj["fp16"]
    print(f"\n[{regime_fixture}] fp32 {a[0]:.4f} -> {a[-1]:.4f}; fp16 {b[0]:.4f} -> {b[-1]:.4f}; max rel gap {np.max(np.abs(a - b) / a):.2e}")
    assert a[-1] < a[0] and b[-1] < b[0]
    assert np.max(np.abs(a - b) / a) < tol


def test_streams_that_must_overlap_are_measured_to():
    """poseidon_amd/streams.py: the ROCm runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues round-robin, so every 7th
    stream created shares the default stream's queue (with 8 queues) — and a pair that shares a queue serialises as soon as it exchanges
    events, which the engine's fork / join pattern does ~130 times per step.  The weight-gradient stream and the gradient exchange's comm
    stream are therefore chosen by MEASURING that pattern (round 6: three spin kernels across a fork / join take two kernel times, not
    three), and a verified stream is shared by every engine of the process.  The census over 20 fresh streams must find both kinds on a box
    with the default 8 queues; whatever it finds, the chosen streams must pass."""
    from poseidon_amd import streams
    from poseidon_amd.streams import independent_stream, overlaps
    main = torch.cuda.current_stream()
    assert not overlaps(main, main)
    pool = [torch.cuda.Stream() for _ in range(20)]
    shared = [i for i, s in enumerate(pool) if not overlaps(main, s)]
    print(f"\nfresh streams that serialise with the main stream under fork / join: {shared} of {len(pool)}")
    assert len(shared) < len(pool)
    side = independent_stream(torch.device(DEV), [main])
    comm = independent_stream(torch.device(DEV), [main, side])
    assert overlaps(main, side) and overlaps(side, main) and overlaps(main, comm) and overlaps(side, comm) and overlaps(comm, side)
    assert independent_stream(torch.device(DEV), [main]) is side            # measured once per process, then shared
    # every engine of the process runs its weight gradients on that one stream
    f, meta = load_fixture("tiny_trained")
    engines = []
    for _ in range(3):
        cfg, model = build(meta, "fp32")
        model(**inputs(cfg, meta)).loss.backward()
        engines.append(model._engine)
    torch.cuda.synchronize()
    assert all(e.side is engines[0].side for e in engines) and overlaps(main, engines[0].side)
    assert len(streams._cache) <= 4


def test_overlapped_gradient_exchange_under_a_one_rank_rccl_group():
    """Data-parallel readiness on the one GPU the driver's run has (reference: `accelerate launch` / DDP, README.md:50-57, train.py:281): a
    1-rank `nccl` (= RCCL) process group, `OverlappedGradAllReducer` attached with the fp32 and then the bf16 wire, three steps each
    (direct, recorded, replayed).  The exchanged gradients must equal the bare run's (fp32 wire: a one-rank mean is the identity — up to the
    order in which the backward's float atomics commit, 5e-6 run to run; bf16 wire: to bfloat16 rounding), and the step must not fall off the hardware-queue cliff DESIGN §7 describes (an RCCL
    communicator's streams sharing a queue with the engine's two: 27.1 vs 21.0 ms) — bound 25 % here, 1-5 % measured."""
    import socket
    import torch.distributed as dist
    from poseidon_amd.dp import OverlappedGradAllReducer
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    pv, t, lab = synth_inputs(16, 4, 4, 128, "smooth")
    kw = dict(pixel_values=pv.to(DEV), time=t.to(DEV), labels=lab.to(DEV))
    cfg, sd, model = _preset_model("B", 128, 4, "fp16")

    def steps(n):
        for _ in range(n):
            model.zero_grad()
            model(**kw).loss.backward()
        torch.cuda.synchronize()

    def timed(n=6):
        steps(2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            model.zero_grad()
            model(**kw).loss.backward()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    steps(3)
    bare_ms = timed()
    bare = model.flat_grads().clone()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        for wire, tol in (("fp32", 1e-4), ("bf16", 4e-3)):      # (fp32 wire: the 16-bit backward's own run-to-run noise, 1.5e-5 measured for this model)
            red = OverlappedGradAllReducer(model, dist, wire=wire)
            red.attach()
            steps(3)
            ms = timed()
            red.finish()
            torch.cuda.synchronize()
            g = model.flat_grads()
            d = float((g - bare).norm() / bare.norm())
            print(f"\n[1-rank RCCL, {wire} wire] step {ms:.2f} ms (bare {bare_ms:.2f}); gradients vs bare run rel-L2 {d:.2e}; "
                  f"{red.bytes_on_wire / 1e6:.0f} MB handed to the collectives")
            assert torch.isfinite(g).all() and d <= tol
            # (the hardware-queue cliff is 1.8x — 21.6 vs 11.4 ms when the side stream shared the default stream's queue, round 6; the bound
            #  leaves room for a busy box: ADVICE r5)
            assert ms < 1.25 * bare_ms + 0.5, (ms, bare_ms)
            red.detach()
    finally:
        dist.destroy_process_group()


def test_native_exchange_through_the_c_abi_on_a_one_rank_communicator():
    """SURVEY §8(b)'s `scot_dp_init / scot_dp_allreduce_bucket / scot_dp_finalize` (csrc/dp.hip; reference: the DDP all-reduce behind
    `accelerate launch`, README.md:50-57, train.py:281) with NO torch process group: rank 0 draws the token, joins a 1-rank RCCL
    communicator, and the overlapped exchange of a Poseidon-B backward runs on it (`backend="native"`): a one-rank mean is the identity on
    the fp32 wire and bfloat16 rounding on the 16-bit wire; the raw entry points sum in place on the stream they are given."""
    from poseidon_amd import ops
    from poseidon_amd.dp import OverlappedGradAllReducer, native_init
    assert ops.dp_world() == 0 and ops.dp_rank() == -1
    x = torch.randn(1 << 20, device=DEV)
    with pytest.raises(Exception):
        ops.dp_allreduce(x)                                  # before init: SCOT_ERR_UNSUPPORTED, loudly
    assert native_init(None) == (1, 0) and ops.dp_world() == 1 and ops.dp_rank() == 0
    try:
        with pytest.raises(Exception):
            ops.dp_init(ops.dp_unique_id(), 0, 1)            # one communicator per process
        ref = x.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.dp_allreduce(x)
            w = x.to(torch.bfloat16)
            w0 = w.clone()
            ops.dp_allreduce(w)
        side.synchronize()
        assert torch.equal(x, ref) and torch.equal(w, w0)
        pv, t, lab = synth_inputs(16, 4, 4, 128, "smooth")
        kw = dict(pixel_values=pv.to(DEV), time=t.to(DEV), labels=lab.to(DEV))
        cfg, sd, model = _preset_model("B", 128, 4, "fp16")

        def steps(n):
            for _ in range(n):
                model.zero_grad()
                model(**kw).loss.backward()
            torch.cuda.synchronize()

        steps(3)
        bare = model.flat_grads().clone()
        for wire, tol in (("fp32", 1e-4), ("bf16", 4e-3)):
            red = OverlappedGradAllReducer(model, None, wire=wire, backend="native")
            assert (red.world, red.rank) == (1, 0)
            red.attach()
            steps(3)                                          # direct, recorded, replayed
            red.finish()
            torch.cuda.synchronize()
            g = model.flat_grads()
            d = float((g - bare).norm() / bare.norm())
            print(f"\n[native 1-rank RCCL, {wire} wire] gradients vs bare run rel-L2 {d:.2e}; {red.bytes_on_wire / 1e6:.0f} MB handed to scot_dp_allreduce_bucket")
            assert torch.isfinite(g).all() and d <= tol and red.bytes_on_wire >= 2.9 * g.numel() * (4 if wire == "fp32" else 2)
            red.detach()
        p0 = model.flat_parameters().clone()
        red.broadcast_parameters(0)
        torch.cuda.synchronize()
        assert torch.equal(model.flat_parameters(), p0)
    finally:
        ops.dp_finalize()
    assert ops.dp_world() == 0
    ops.dp_finalize()                                        # idempotent


def test_short_training_run_fp16_tracks_fp32_on_poseidon_T():
    """The same check on a BASELINE model (Poseidon-T, 128 x 128 x 4, batch 4, trained-like parameters): 6 fused-AdamW steps in fp32 and in
    fp16 from the same state — the reference trains in fp32 (train.py:277-323), the headline number is measured in fp16."""
    from scOT.trainer import FusedAdamW
    pv, t, lab = synth_inputs(4, 4, 4, 128, "smooth")
    kw = dict(pixel_values=pv.to(DEV), time=t.to(DEV), labels=lab.to(DEV))
    traj = {}
    for compute in ("fp32", "fp16"):
        cfg, sd, model = _preset_model("T", 128, 4, compute)
        opt = FusedAdamW(model, lr=5e-5, weight_decay=0.01, max_grad_norm=5.0)
        losses = []
        for _ in range(6):
            opt.zero_grad()
            out = model(**kw)
            out.loss.backward()
            opt.step()
            losses.append(float(out.loss.detach()))
        torch.cuda.synchronize()
        if compute == "fp16":
            assert int(model._engine.grad_overflow) == 0
        traj[compute] = np.array(losses)
        del model, opt
    a, b = traj["fp32"], traj["fp16"]
    print(f"\n[Poseidon-T] fp32 {a[0]:.4f} -> {a[-1]:.4f}; fp16 {b[0]:.4f} -> {b[-1]:.4f}; max rel gap {np.max(np.abs(a - b) / a):.2e}")
    assert a[-1] < a[0] and b[-1] < b[0]
    assert np.max(np.abs(a - b) / a) < 2e-3          # (measured 2.4e-4: 15.337 -> 9.132 in both modes)


def test_training_with_lazy_zero_grad_tracks_the_eager_fill():
    """Six fused-AdamW steps of Poseidon-T (fp16) with `opt.zero_grad(overlap=True)` — the training loops' form: lazy fill, first writers store,
    un-scale folded into the stores — against the same six steps with the eager fill: same loss trajectory and same final weights up to the
    16-bit backward's own run-to-run noise; the optimizer's clip / skip logic sees finite norms throughout."""
    from scOT.trainer import FusedAdamW
    pv, t, lab = synth_inputs(4, 4, 4, 128, "smooth")
    kw = dict(pixel_values=pv.to(DEV), time=t.to(DEV), labels=lab.to(DEV))
    traj, final = {}, {}
    for lazy in (False, True):
        cfg, sd, model = _preset_model("T", 128, 4, "fp16")
        opt = FusedAdamW(model, lr=5e-5, weight_decay=0.01, max_grad_norm=5.0)
        losses = []
        for _ in range(6):
            opt.zero_grad(overlap=lazy)
            out = model(**kw)
            out.loss.backward()
            opt.step()
            losses.append(float(out.loss.detach()))
        torch.cuda.synchronize()
        assert int(model._engine.grad_overflow) == 0 and opt.skipped_steps() == 0
        assert bool(model._engine.lazy_grads) is False                      # consumed by the backward
        traj[lazy], final[lazy] = np.array(losses), model.flat_parameters().clone()
        del model, opt
    gap = float(np.max(np.abs(traj[True] - traj[False]) / traj[False]))
    dw = float((final[True] - final[False]).norm() / final[False].norm())
    print(f"\n[lazy vs eager zero_grad, Poseidon-T fp16] loss {traj[False][0]:.4f} -> {traj[False][-1]:.4f} / {traj[True][-1]:.4f}; max rel gap {gap:.2e}; weights rel-L2 {dw:.2e}")
    assert traj[True][-1] < traj[True][0] and gap < 1e-3 and dw < 1e-4


@pytest.mark.parametrize("compute", ["fp16", "fp32"])
def test_grad_ranges_are_final_when_announced(compute):
    """The data-parallel hook on the real streams: `on_grads_final(prefix)` runs with the side stream current (behind the range's
    weight gradients and, in fp16 mode, its un-scale) while the main chain goes on.  Whatever the callback enqueues on its current
    stream must see the range's FINAL values: a snapshot taken there equals the gradient arena after the backward, for every
    range, on recorded and replayed steps alike; the ranges arrive in the documented order and cover every parameter."""
    from poseidon_amd.dp import backward_order_groups, group_ranges
    f, meta = load_fixture("tiny_trained")
    cfg, model = build(meta, compute)
    kw = inputs(cfg, meta)
    model(**kw).loss.backward()
    eng, arena = model._engine, model._arena
    snap = torch.zeros_like(arena.grad)       # (alignment gaps of the arena are never announced and stay zero in both)
    seen = []

    def on_final(prefix):
        seen.append(prefix)
        for _, lo, hi in group_ranges(arena, [prefix]):
            snap[lo:hi].copy_(arena.grad[lo:hi])          # on the stream the engine made current for the announcement
    eng.on_grads_final = on_final
    eng.reset_tapes()
    try:
        for step in range(4):                             # direct, recording and replayed steps
            seen.clear()
            snap.zero_()
            model.zero_grad()
            model(**kw).loss.backward()
            torch.cuda.synchronize()
            assert torch.equal(snap, arena.grad), (step, float((snap - arena.grad).abs().max()))
            assert seen == [g for g in backward_order_groups(cfg) if g in seen] and len(seen) >= 5
    finally:
        eng.on_grads_final = None
        eng.reset_tapes()


def test_trainer_on_device_resident_dataset():
    """The driver with the reference trainer's surface on the GPU: fused arena-wide AdamW (reference's four parameter groups), cosine
    schedule, batches gathered from HBM-resident trajectories, evaluation through the reference-style sample dicts."""
    from scOT.problems.base import get_dataset
    from scOT.trainer import FusedAdamW, Trainer, TrainingArguments
    rng = np.random.default_rng(0)
    x = rng.standard_normal((12, 21, 5, 32, 32)).astype(np.float32)
    rd = {"data": np.cumsum(0.2 * x, axis=1).astype(np.float32)}            # trajectories with some temporal structure
    kw = dict(reader=rd, n_max=12, n_val=3, n_test=3, max_num_time_steps=3, time_step_size=2)
    train = get_dataset("fluids.compressible.Riemann", which="train", num_trajectories=6, **kw)
    val = get_dataset("fluids.compressible.Riemann", which="val", num_trajectories=6, **kw)
    for d in (train, val):
        d.resolution = 32
    f, meta = load_fixture("tiny_trained")
    cfg, model = build(meta, "fp16")
    args = TrainingArguments(per_device_train_batch_size=8, per_device_eval_batch_size=8, num_train_epochs=2, learning_rate=1e-3,
                             learning_rate_embedding_recovery=5e-4, weight_decay=0.01, lr_scheduler_type="cosine", warmup_ratio=0.1,
                             logging_steps=1, max_grad_norm=5.0)
    tr = Trainer(model=model, args=args, train_dataset=train.to_device(DEV), eval_dataset=val,
                 compute_metrics=lambda p: {"l1": float(np.abs(p.predictions - p.label_ids).mean())})
    before = tr.evaluate()
    out = tr.train()
    after = tr.evaluate()
    assert isinstance(tr.optimizer, FusedAdamW) and len(tr.optimizer.param_groups) == 3
    steps = 2 * math.ceil(len(train) / 8)
    assert out.global_step == steps and len([h for h in tr.state["log_history"] if "grad_norm" in h]) == steps
    assert after["eval_loss"] < before["eval_loss"] and after["eval_l1"] < before["eval_l1"] and int(model._engine.grad_overflow) == 0
    tr.set_ar_steps([1, 1])
    p = tr.predict(val, metric_key_prefix="")
    assert p.predictions.shape == (len(val), 4, 32, 32) and np.isfinite(p.metrics["_loss"])


@pytest.mark.parametrize("compute,tol", [("fp32", 3e-5), ("fp16", 2e-2)])     # (fp16 on the 16-wide toy model: 6.5e-3 measured at the deepest stage)
def test_output_attentions_match_reference(compute, tol):
    """`output_attentions=True` against the real reference (tests/golden/make_attentions_fixture.py): one probability tensor per stage
    (its last block), decoder stages first; the `return_dict=False` tuple carries them behind each stack's hidden states."""
    f, meta = load_fixture("tiny_attentions")
    cfg, model = build(meta, compute)
    model.eval()
    kw = inputs(cfg, meta)
    with torch.no_grad():
        out = model(**kw, output_attentions=True)
        tup = model(**kw, output_attentions=True, output_hidden_states=True, return_dict=False)
    assert len(out.attentions) == meta["n_attn"]
    for i, a in enumerate(out.attentions):
        ref = torch.from_numpy(f[f"attn:{i}"]).to(a.device)
        assert tuple(a.shape) == tuple(ref.shape) and float((a - ref).abs().max()) < tol, (i, float((a - ref).abs().max()))
    assert [(-1 if torch.is_tensor(x) else len(x)) for x in tup] == f["tuple_layout"].tolist()


def test_output_attentions_on_16x16_windows():
    """Poseidon-T (16x16 windows: the fast-path kernels' log-sum-exp, shifted windows at stage 0): every row of every returned
    probability tensor sums to 1 in the default fp16 mode."""
    f, meta = load_fixture("poseidonT_trained")
    cfg, model = build(meta, "fp16")
    model.eval()
    with torch.no_grad():
        out = model(**inputs(cfg, meta), output_attentions=True)
    assert len(out.attentions) == 2 * len(cfg.depths)
    assert tuple(out.attentions[-len(cfg.depths)].shape) == (meta["batch"] * 4, cfg.num_heads[0], 256, 256)     # encoder stage 0: 2x2 windows of 16x16
    for a in out.attentions:
        assert float((a.sum(-1) - 1).abs().max()) < 2e-2 and float(a.min()) >= 0
    assert rel_l2(out.output.detach().cpu().numpy(), f["output"]) < 1e-3


def test_overlapped_gradient_fill_is_ordered_before_the_backward():
    """ScOT.zero_grad(overlap=True): the arena's fill runs on the weight-gradient stream beside the next forward; the backward's
    gradient writers (both streams) wait for it.  Same gradients as with the in-order fill, on direct, recorded and replayed steps,
    starting from an arena full of garbage each time."""
    cfg = ScOTConfig(image_size=32, patch_size=4, num_channels=4, num_out_channels=4, embed_dim=16, depths=[2, 2], num_heads=[1, 2],
                     skip_connections=[1, 0], window_size=4, mlp_ratio=4.0, p=1, channel_slice_list_normalized_loss=[0, 1, 3, 4],
                     drop_path_rate=0.0, use_conditioning=True)
    sd = synth_state_dict(param_shapes(cfg), "trained")
    pv, t, lab = synth_inputs(2, 4, 4, 32, "smooth")
    pv, t, lab = pv.to(DEV), t.to(DEV), lab.to(DEV)
    grads = {}
    for overlap in (False, True):
        model = ScOT(cfg, compute="fp32")
        model.load_state_dict(sd)
        model = model.to(DEV)
        for step in range(4):
            if model._arena is not None:
                model._arena.grad.fill_(float("nan") if step % 2 else 7.0)      # what a forgotten fill would leave behind
            model.zero_grad(overlap=overlap)
            out = model(pixel_values=pv, time=t, labels=lab)
            out.loss.backward()
        torch.cuda.synchronize()
        grads[overlap] = model._arena.grad.clone()
    assert bool(torch.isfinite(grads[True]).all())
    d = float((grads[True] - grads[False]).norm() / grads[False].norm())      # (not bit-equal: float atomics commit in any order)
    assert d < 5e-6, d


@pytest.mark.parametrize("tag,size,channels,batch", [("B", 128, 4, 8), ("T", 128, 4, 4)])
@pytest.mark.parametrize("compute", ["fp16", "bf16"])
def test_lazy_zero_grad_equals_the_eager_fill(tag, size, channels, batch, compute):
    """Round 6: ScOT.zero_grad(overlap=True) fills only the part of the gradient arena that is accumulated into; the Linear weights of the
    ScOTLayers (95 % of the bytes) are STORED by the next backward's weight-gradient kernels, fp16 un-scale included.  Starting from NaN in
    the unfilled part every time: direct launches, the recorded step and its replays give the gradients of the eager form (whole arena
    filled, accumulated into, un-scaled); a second backward without zero_grad accumulates (the recorded `add` variant); every tensor is
    finite, i.e. nothing in the stored set was left without its first writer."""
    cfg, sd, model = _preset_model(tag, size, channels, compute)
    pv, t, lab = synth_inputs(batch, channels, channels, size, "smooth")
    kw = dict(pixel_values=pv.to(DEV), time=t.to(DEV), labels=lab.to(DEV))
    model(**kw).loss.backward()
    eng, ar = model._engine, model._arena
    assert eng._small_chunks is not None
    small = torch.zeros(ar.size, dtype=torch.bool, device=DEV)
    for o, n in eng._small_chunks[0].tolist():
        small[o:o + n] = True
    frac = float(small.float().mean())
    assert frac < 0.12, frac                      # > 88 % of the arena is never filled (Poseidon-B: 95 %)
    model.zero_grad()
    model(**kw).loss.backward()
    torch.cuda.synchronize()
    ref = ar.grad.clone()
    assert bool(torch.isfinite(ref).all())
    # the 16-bit backward is not bit-reproducible (fp32 atomics commit in any order, an ulp upstream flips 16-bit roundings downstream): the
    # floor is what two EAGER steps differ by — 1.5e-5 (B) .. 1e-4 (T) in fp16, ~8x that in bf16 on this box; a missing 1/S, a stale or
    # an unfilled tensor is an O(1) error
    model.zero_grad()
    model(**kw).loss.backward()
    torch.cuda.synchronize()
    floor = float((ar.grad - ref).norm() / ref.norm())
    bound = 4.0 * floor + 2e-5
    print(f"\n[lazy zero-grad {tag} {compute}] eager-vs-eager floor {floor:.2e}, bound {bound:.2e}; filled fraction of the arena {frac:.3f}")
    assert bound < 2e-2
    for step in range(4):                          # direct / recorded / replayed / replayed
        ar.grad[~small] = float("nan")
        ar.grad[small] = 3.0
        model.zero_grad(overlap=True)
        assert eng.lazy_grads
        model(**kw).loss.backward()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(ar.grad).all()), step
        d = float((ar.grad - ref).norm() / ref.norm())
        assert d < bound, (step, d, floor)
        for p_ in sorted(eng._big_ptrs)[::7]:
            o = (p_ - ar.grad.data_ptr()) // 4
            a_, b_ = ar.grad[o:o + 4096], ref[o:o + 4096]
            assert float((a_ - b_).norm()) <= 0.25 * float(b_.norm()) + 1e-12, (step, o)      # (a stored tensor is its eager value, not 2^k times it or stale)
    model(**kw).loss.backward()                    # accumulation window: no zero_grad
    torch.cuda.synchronize()
    d = float((ar.grad - 2 * ref).norm() / (2 * ref).norm())
    assert d < bound, (d, floor)
    assert eng.grad_overflow is None or int(eng.grad_overflow) == 0
    ents = [e for e in eng._taped.values() if e.get("state") == "ready"]
    assert ents and any(set(e["bwd"]) == {True, False} for e in ents)


# ----------------------------------------------------------------------------------------------- the TIMED batch sizes
def _preset_model(tag, size, channels, compute, regime="trained", engine_options=None):
    from poseidon_amd.config import preset
    cfg = preset(tag, image_size=size, num_channels=channels, num_out_channels=channels,
                 channel_slice_list_normalized_loss=[0, 1, channels - 1, channels])
    sd = synth_state_dict(param_shapes(cfg), regime)
    model = ScOT(cfg, compute=compute, engine_options=engine_options)
    model.load_state_dict(sd)
    return cfg, sd, model.to(DEV)


# BASELINE configs 3 / 2 / 5 / 4 at their timed batches (config 4 — Poseidon-L, train.py:62-71 — both as the per-device batch 128 and as the
# 8-way share of a global 128)
TIMED = [("B", 128, 4, 64), ("T", 128, 4, 32), ("B", 256, 4, 32), ("L", 128, 5, 128), ("L", 128, 5, 16)]


@pytest.mark.parametrize("tag,size,channels,batch", TIMED)
@pytest.mark.parametrize("compute", ["fp32", "fp16"])
def test_timed_batch_matches_the_batch1_path(tag, size, channels, batch, compute):
    """BASELINE configs 3 / 2 / 5 / 4 at the batch sizes bench.py times: the large row counts select launch policies no batch-1 fixture
    reaches (128-row block tails, grouped / recomputing weight gradients, direct-to-LDS and four-register-set GEMMs, the 128 x 128-tile
    GEMMs and grouped weight gradients — whose fp32 sums run in another order than the batch-1 path's 64 x 64 tiles: agreement to summation
    order, not bit for bit —, the XCD-local attention grid).  Samples are independent, so (i) prediction[i] of the batch must equal the prediction of sample i alone, and (ii)
    the batch's parameter gradients must equal the sum of the per-sample gradients weighted as the relative loss weights them — checked
    through the loss and the full gradients of a few samples' worth (a batch of 3 against 3 batches of 1).
    (fused_min_rows=0: the single samples stay on the fused layer tails, the kernel family of the timed batch — round 6's policy hands batches
    below 4096 token rows to the layer-by-layer GEMMs, whose 16-bit rounding points differ: 4e-4 instead of < 2e-4 — so that what is compared is
    one kernel family at two sizes; the layer-by-layer path at small batches has its own references: the oracle test below, the fp32-path test.)"""
    cfg, sd, model = _preset_model(tag, size, channels, compute, engine_options={"fused_min_rows": 0})
    pv, t, lab = synth_inputs(batch, channels, channels, size, "smooth")
    pv, t, lab = pv.to(DEV), t.to(DEV), lab.to(DEV)
    out = model(pixel_values=pv, time=t, labels=lab)
    out.loss.backward()
    full = out.output.detach().clone()
    torch.cuda.synchronize()
    assert torch.isfinite(full).all() and torch.isfinite(model.flat_grads()).all()
    worst = 0.0
    for i in (0, batch // 3, batch - 1):
        one = model(pixel_values=pv[i:i + 1].contiguous(), time=t[i:i + 1].contiguous(), labels=lab[i:i + 1].contiguous()).output.detach()
        worst = max(worst, rel_l2(one.cpu().numpy(), full[i:i + 1].cpu().numpy()))
    print(f"\n[Poseidon-{tag} {size}^2 batch {batch} {compute}] prediction[i] vs sample i alone: max rel-L2 {worst:.2e}")
    # same products, same rounding points; only the tile shapes (hence fp32 summation order) may differ
    assert worst < (2e-6 if compute == "fp32" else 2e-4)


@pytest.mark.parametrize("compute", ["fp32", "fp16"])
def test_config3_batch8_against_the_oracle(compute):
    """Poseidon-B (BASELINE config 3's model) at a batch the CPU oracle can hold (8): output, loss and every parameter
    gradient of the HIP path against oracle/scot_cpu.py on the same closed-form parameters and inputs."""
    from oracle import scot_cpu
    cfg, sd, model = _preset_model("B", 128, 4, compute)
    pv, t, lab = synth_inputs(8, 4, 4, 128, "smooth")
    out = model(pixel_values=pv.to(DEV), time=t.to(DEV), labels=lab.to(DEV))
    out.loss.backward()
    torch.cuda.synchronize()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss, pred = scot_cpu.scot_forward(ref, cfg, pv, t, lab)
    loss.backward()
    e_out = rel_l2(out.output.detach().cpu().numpy(), pred.detach().numpy())
    e_loss = abs(float(out.loss) - float(loss)) / abs(float(loss))
    num = den = 0.0
    worst = (0.0, "")
    for k, p in model.named_parameters():
        r = ref[k].grad
        if r is None:
            continue
        err, nr = float((p.grad.detach().cpu().double() - r.double()).norm()), float(r.double().norm())
        if nr > 0 and err / nr > worst[0]:
            worst = (err / nr, k)
        num += err ** 2
        den += nr ** 2
    g = (num / den) ** 0.5
    print(f"\n[Poseidon-B batch 8 {compute} vs oracle] out rel-L2 {e_out:.2e} loss rel {e_loss:.2e} grads global rel-L2 {g:.2e} worst {worst}")
    if compute == "fp32":
        assert e_out < 1e-5 + 5e-6 and e_loss < 2e-5 and g < 1e-3
    else:
        assert e_out < 1e-3 and e_loss < 1e-3 and g < 2.5e-3      # (measured 7.1e-4 / 1.6e-4 / 8.2e-4)
        assert int(model._engine.grad_overflow) == 0


@pytest.mark.parametrize("tag,size,channels,batch", TIMED)
def test_timed_batch_gradients_fp16_vs_fp32_path(tag, size, channels, batch):
    """The backward at the timed sizes: the fp16 step (lean 128-row tails, recomputing fc1 / fc2 weight gradients, grouped weight
    gradients over 65536 tokens, the transposed-copy data gradients) against the fp32 mode of the same engine AT THE SAME BATCH — a
    different set of kernels (exact fp32 MFMA, layer-by-layer launches) that the oracle pins to 1e-6 at the sizes a CPU can hold
    (test_config3_batch8_against_the_oracle, the fixtures).  Bounds: those of the fp16 mode against the reference fixtures."""
    pv, t, lab = synth_inputs(batch, channels, channels, size, "smooth")
    kw = dict(pixel_values=pv.to(DEV), time=t.to(DEV), labels=lab.to(DEV))
    res = {}
    for compute in ("fp32", "fp16"):
        cfg, sd, model = _preset_model(tag, size, channels, compute)
        for rep in range(3):        # direct launches, the recorded step, its replay: the third is what bench.py times
            model.zero_grad()
            out = model(**kw)
            out.loss.backward()
        torch.cuda.synchronize()
        if compute == "fp16":
            assert int(model._engine.grad_overflow) == 0
        res[compute] = (out.output.detach().clone(), float(out.loss), {k: p.grad.detach().clone() for k, p in model.named_parameters()})
        del model
    e_out = rel_l2(res["fp16"][0].cpu().numpy(), res["fp32"][0].cpu().numpy())
    e_loss = abs(res["fp16"][1] - res["fp32"][1]) / abs(res["fp32"][1])
    num = den = 0.0
    worst = (0.0, "")
    for k, r in res["fp32"][2].items():
        if "logit_scale" in k:
            continue
        err, nr = float((res["fp16"][2][k].double() - r.double()).norm()), float(r.double().norm())
        if nr > 1e-7 * max(1.0, den ** 0.5) and err / nr > worst[0]:
            worst = (err / nr, k)
        num += err ** 2
        den += nr ** 2
    g = (num / den) ** 0.5
    print(f"\n[Poseidon-{tag} {size}^2 batch {batch}] fp16 vs fp32 path: out rel-L2 {e_out:.2e} loss rel {e_loss:.2e} grads global rel-L2 {g:.2e} worst {worst}")
    # measured (round 5): out 6.1e-4 .. 8.9e-4, loss 1.3e-5 .. 1.6e-4, gradients 4.5e-4 .. 8.0e-4 over the five timed configurations
    assert e_out < 1e-3 and e_loss < 1e-3 and g < 2.5e-3


def test_inference_forwards_are_taped_and_replay_like_direct_launches():
    """Inference forwards of one signature are recorded (call 2) and replayed (call 3+) — a rollout is hundreds of them.  Different inputs
    per call; against a model that never tapes; interleaved with a training step of another signature (its own tape, its own hidden
    states)."""
    f, meta = load_fixture("tiny_trained")
    cfg, model = build(meta, "fp16")
    _, plain = build(meta, "fp16")
    kw = inputs(cfg, meta)
    with torch.no_grad():
        plain(pixel_values=kw["pixel_values"], time=kw["time"])      # (creates the engine: a first call only warms a signature)
    plain._engine.tape_mode = False
    g = torch.Generator(device="cpu").manual_seed(5)
    held = None
    for call in range(5):
        pv = (kw["pixel_values"] + 0.1 * torch.randn(kw["pixel_values"].shape, generator=g).to(DEV)).contiguous()
        model.eval()
        plain.eval()
        with torch.no_grad():
            a = model(pixel_values=pv, time=kw["time"], output_hidden_states=True)
            b = plain(pixel_values=pv, time=kw["time"], output_hidden_states=True)
        model.train()
        assert torch.equal(a.output, b.output), call
        for ha, hb in zip(a.hidden_states, b.hidden_states):
            assert torch.equal(ha, hb), call
        # the hidden states handed out by an EARLIER call are the caller's own (the reference returns fresh tensors): a replay
        # of the recorded step must not have changed them
        if held is not None:
            for h_old, h_copy in zip(held[0], held[1]):
                assert torch.equal(h_old, h_copy), call
        held = (a.hidden_states, [h.clone() for h in a.hidden_states])
        if call == 2:       # a training step in between: another signature, another recorded step
            out = model(**kw)
            out.loss.backward()
    states = [e["state"] for e in model._engine._taped.values()]
    assert "ready" in states
