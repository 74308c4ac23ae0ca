"""The ENGINE on the CPU: `poseidon_amd.engine` (the host program: stage plan, padding, window shifts, skip wiring, loss,
backward order) driving every kernel of libscot_emu.so — the kernel sources compiled for the host through tests/hipemu — on
CPU tensors, checked against the golden vectors of the real reference and against the oracle.  Same bounds as the GPU tests
(tests/test_model_gpu.py); the numbers it produces equal the ones measured on the MI355X (tiny fp32 1.8e-6 / gradients 1.2e-5).
Test infrastructure: ScOT.forward itself refuses CPU tensors, so the tests call the engine below that guard."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "hipemu"))
from conftest import load_fixture, rel_l2  # noqa: E402
from poseidon_amd import engine as engine_mod, ops  # noqa: E402
from poseidon_amd.config import ScOTConfig  # noqa: E402
from poseidon_amd.geometry import param_shapes  # noqa: E402
from poseidon_amd.synth import apply_obstacle, synth_inputs, synth_obstacle_mask, synth_state_dict  # noqa: E402

FULL = os.environ.get("SCOT_EMU_FULL") == "1"


@pytest.fixture()
def emu(monkeypatch):
    import emu_session
    lib = emu_session.load_emu()
    emu_session.patch_ops(monkeypatch, lib)
    monkeypatch.setenv("SCOT_SIDE_STREAM", "0")     # HIP streams / events do not exist here: one in-order "stream"
    monkeypatch.setenv("SCOT_TAPE", "0")
    # the emulated models are tiny: without this the engine's policy (fused layer tails from 4096 token rows) would keep them off the
    # fused kernels these tests are here to drive through the engine
    monkeypatch.setitem(engine_mod.ENGINE_OPTIONS, "fused_min_rows", 0)
    return lib


def run_engine(cfg, sd, pv, t, lab, mask, compute, grads=True, drop_masks=None, bool_masked_pos=None):
    from scOT.model import ScOT
    model = ScOT(cfg, compute=compute, use_mask_token=bool_masked_pos is not None)
    model.load_state_dict(sd)
    model._ensure_arena(torch.device("cpu"))
    if drop_masks is not None:
        model._engine.drop_path_masks = drop_masks
    loss, pred, tape = model._engine.forward(pv, t if cfg.use_conditioning else None, lab, mask, train=grads,
                                             bool_masked_pos=bool_masked_pos)
    if grads:
        model._prepare_grads()
        model._engine.backward(tape, torch.ones(1), None)
    return model, loss, pred


def fixture_inputs(meta, cfg):
    size = meta.get("size", cfg.image_size)
    pv, t, lab = synth_inputs(meta["batch"], cfg.num_channels, cfg.num_out_channels, size, meta["kind"])
    pm = None
    if meta.get("with_mask") == "obstacle":
        pm = synth_obstacle_mask(meta["batch"], size)
        pv, lab = apply_obstacle(pv, lab, pm)
    elif meta.get("with_mask"):
        pm = torch.zeros(meta["batch"], cfg.num_out_channels, dtype=torch.bool)
        pm[:, -1] = True
    return pv, t, lab, pm


def grads_global(model, f):
    num = den = 0.0
    for k, p in model.named_parameters():
        if "grad:" + k in f.files:
            ref = f["grad:" + k].astype(np.float64)
            num += float(((p.grad.numpy().astype(np.float64) - ref) ** 2).sum())
            den += float((ref ** 2).sum())
    return (num / max(den, 1e-300)) ** 0.5


TINY = ["tiny_trained", "tiny_hf", "tiny_odd", "tiny_shift3", "tiny_nocond_p2", "tiny_learnres_mask", "tiny_obstacle_mask", "tiny_abspos"]


@pytest.mark.parametrize("name", TINY)
def test_engine_fp32_vs_reference_fixture(emu, name):
    f, meta = load_fixture(name)
    cfg = ScOTConfig(**meta["cfg"])
    pv, t, lab, pm = fixture_inputs(meta, cfg)
    model, loss, pred = run_engine(cfg, synth_state_dict(param_shapes(cfg), meta["regime"]), pv, t, lab, pm, "fp32")
    assert rel_l2(pred.numpy(), f["output"]) < 1e-5 + 5e-6
    assert abs(float(loss) - float(f["loss"])) < 2e-5 * abs(float(f["loss"]))
    assert grads_global(model, f) < 1e-4


def test_engine_mask_tokens(emu):
    """use_mask_token / bool_masked_pos (reference model.py:323-327, 353-359) against the real reference's fixture."""
    from poseidon_amd.synth import synth_token_mask
    f, meta = load_fixture("tiny_masktoken")
    cfg = ScOTConfig(**meta["cfg"])
    pv, t, lab, pm = fixture_inputs(meta, cfg)
    bmp = synth_token_mask(meta["batch"], (cfg.image_size // cfg.patch_size) ** 2)
    sd = synth_state_dict(param_shapes(cfg, use_mask_token=True), meta["regime"])
    model, loss, pred = run_engine(cfg, sd, pv, t, lab, pm, "fp32", bool_masked_pos=bmp)
    assert rel_l2(pred.numpy(), f["output"]) < 1e-5 + 5e-6
    assert abs(float(loss) - float(f["loss"])) < 2e-5 * abs(float(f["loss"]))
    assert grads_global(model, f) < 1e-4
    g = dict(model.named_parameters())["embeddings.mask_token"].grad
    assert rel_l2(g.numpy(), f["grad:embeddings.mask_token"]) < 1e-4


@pytest.mark.parametrize("compute,tol_out,tol_grad", [("bf16", 5e-2, 0.7), ("bf16x3", 1e-4, 2e-3), ("fp16", 5e-3, 5e-2)])
def test_engine_reduced_precision_modes(emu, compute, tol_out, tol_grad):
    """The 16-dim tiny model is ~4x more sensitive to operand rounding than Poseidon-T/B (contractions of length 16): fp16 gives
    3e-3 here exactly as tools/probes/precision_sim.py predicts (bf16: 2e-2), and 7e-4..8e-4 on T/B, where the north star's 1e-3
    is asserted (tests/test_model_gpu.py)."""
    f, meta = load_fixture("tiny_trained")
    cfg = ScOTConfig(**meta["cfg"])
    pv, t, lab, pm = fixture_inputs(meta, cfg)
    model, loss, pred = run_engine(cfg, synth_state_dict(param_shapes(cfg), meta["regime"]), pv, t, lab, pm, compute)
    print(f"\n[{compute}] out {rel_l2(pred.numpy(), f['output']):.2e} grads {grads_global(model, f):.2e}")
    assert rel_l2(pred.numpy(), f["output"]) < tol_out
    assert grads_global(model, f) < tol_grad
    if compute == "fp16":   # the backward ran under the power-of-two gradient scale and came back exactly, without overflow
        assert model._engine.scale_grads and int(model._engine.grad_overflow) == 0


def test_engine_fp16_layer_scale_branch_gradients(emu, monkeypatch):
    """HF-init regime (ConvNeXt layer scale 1e-6): the skip blocks' branch gradients sit ~2^-20 below the rest and flush to zero in
    binary16 under the one global gradient scale; with the device-side local power of two (scot_pow2_rescale / colscale_dev / axpy_dev)
    every parameter of those blocks gets its gradient — also when the arena is accumulated into twice."""
    f, meta = load_fixture("tiny_hf")
    cfg = ScOTConfig(**meta["cfg"])
    pv, t, lab, pm = fixture_inputs(meta, cfg)
    sd = synth_state_dict(param_shapes(cfg), meta["regime"])

    def branch_errors(model):
        out = {}
        for k, p in model.named_parameters():
            if k.startswith("residual_blocks.") and not k.endswith(".weight") or k.endswith(("dwconv.weight", "pwconv1.weight", "pwconv2.weight")):
                if k.startswith("residual_blocks.") and "grad:" + k in f.files and np.linalg.norm(f["grad:" + k]) > 0:
                    out[k] = rel_l2(p.grad.numpy(), f["grad:" + k])
        return out
    model, loss, pred = run_engine(cfg, sd, pv, t, lab, pm, "fp16")
    assert model._engine._ls and int(model._engine.grad_overflow) == 0
    errs = branch_errors(model)
    assert len(errs) >= 8 and max(errs.values()) < 5e-2, errs
    g1 = model._arena.grad.clone()
    eng = model._engine
    _, _, tp = eng.forward(pv, t, lab, pm, train=True)          # accumulate a second, identical backward: exactly 2x
    eng.backward(tp, torch.ones(1), None)
    assert rel_l2(model._arena.grad.numpy(), 2.0 * g1.numpy()) < 1e-6
    monkeypatch.setitem(engine_mod.ENGINE_OPTIONS, "ls_rescale", False)                  # what the rescale is for: without it those gradients are lost
    model0, _, _ = run_engine(cfg, sd, pv, t, lab, pm, "fp16")
    assert not model0._engine._ls and max(branch_errors(model0).values()) > 0.5


def test_engine_fp16_gradient_scale_accumulates(emu):
    """fp16 build: two backwards into the same gradient arena (gradient accumulation) — the second one finds a non-zero arena,
    brings it to the backward's scale first and both contributions come back at scale 1: grads == 2 x the single-step grads."""
    f, meta = load_fixture("tiny_trained")
    cfg = ScOTConfig(**meta["cfg"])
    pv, t, lab, pm = fixture_inputs(meta, cfg)
    from scOT.model import ScOT
    model = ScOT(cfg, compute="fp16")
    model.load_state_dict(synth_state_dict(param_shapes(cfg), meta["regime"]))
    model._ensure_arena(torch.device("cpu"))
    eng = model._engine
    model._prepare_grads()
    assert eng.grads_are_zero
    _, _, tp = eng.forward(pv, t, lab, pm, train=True)
    eng.backward(tp, torch.ones(1), None)
    g1 = model._arena.grad.clone()
    assert not eng.grads_are_zero
    _, _, tp = eng.forward(pv, t, lab, pm, train=True)
    eng.backward(tp, torch.ones(1), None)
    assert rel_l2(model._arena.grad.numpy(), 2.0 * g1.numpy()) < 1e-6
    assert grads_global(model, f) > 0.5          # i.e. really 2x, not 1x
    assert int(eng.grad_overflow) == 0


@pytest.mark.parametrize("compute", ["fp16", "bf16"])
def test_engine_lazy_zero_grad_stores_first_writers(emu, compute):
    """Round 6: after ScOT.zero_grad(lazy=True) the Linear weights of the ScOTLayers are NOT cleared (poisoned with NaN here): their first
    writers in the next backward store acc / S, the rest of the arena was zero-filled and is un-scaled piecewise — the gradients equal
    those of the eager form (full fill, accumulate, un-scale of the whole arena: the `lazy_grads=False` engine option's semantics reproduced by zero_grad()),
    a second backward without zero_grad accumulates onto them (x 2), and no tensor outside the stored set is left unfilled.  C = 96 / 192 with
    16x16 windows (lean tail: scot_wgrad_mlp + grouped gradients) and a ragged tiny model (grouped / single-problem launches)."""
    from scOT.model import ScOT
    cfgs = [ScOTConfig(image_size=64, patch_size=4, num_channels=4, num_out_channels=4, embed_dim=96, depths=[2, 1], num_heads=[3, 6],
                       skip_connections=[1, 0], window_size=16, mlp_ratio=4.0, qkv_bias=True, drop_path_rate=0.0, hidden_act="gelu", p=1,
                       channel_slice_list_normalized_loss=[0, 1, 3, 4], residual_model="convnext", use_conditioning=True, learn_residual=False),
            ScOTConfig(**load_fixture("tiny_odd")[1]["cfg"])]
    for cfg in cfgs:
        if (cfg.embed_dim == 96) != (compute == "fp16"):
            continue            # (the 96-wide model in fp16, the ragged tiny one in bf16: half the emulation time, both code paths in both formats' kernels)
        sd = synth_state_dict(param_shapes(cfg), "trained")
        if cfg.embed_dim == 96:
            pv, t, lab = synth_inputs(1, cfg.num_channels, cfg.num_out_channels, cfg.image_size, "smooth")
        else:
            pv, t, lab, _ = fixture_inputs(load_fixture("tiny_odd")[1], cfg)
        model = ScOT(cfg, compute=compute)
        model.load_state_dict(sd)
        model._ensure_arena(torch.device("cpu"))
        eng, ar = model._engine, model._arena
        assert eng._small_chunks is not None and eng._big_ptrs
        covered = sum(n for _, n in eng._small_chunks[0].tolist())
        assert 0 < covered < ar.size

        def step():
            _, _, tp = eng.forward(pv, t, lab, None, train=True)
            model._prepare_grads()
            eng.backward(tp, torch.ones(1), None)
            return ar.grad.clone()
        model.zero_grad()                       # eager: the whole arena filled
        assert not eng.lazy_grads
        g_eager = step()
        assert torch.isfinite(g_eager).all()
        small = torch.zeros(ar.size, dtype=torch.bool)
        for o, n in eng._small_chunks[0].tolist():
            small[o:o + n] = True
        ar.grad[~small] = float("nan")          # what a lazy zero_grad leaves behind could be anything
        ar.grad[small] = 123.0
        model.zero_grad(lazy=True)
        assert eng.lazy_grads and eng.grads_are_zero and float(ar.grad[small].abs().max()) == 0.0
        g_lazy = step()
        assert not eng.lazy_grads
        assert torch.isfinite(g_lazy).all()
        assert rel_l2(g_lazy.numpy(), g_eager.numpy()) < 1e-6, cfg.embed_dim
        g2 = step()                             # accumulation window: no zero_grad in between
        assert rel_l2(g2.numpy(), 2.0 * g_eager.numpy()) < 1e-6
        assert eng.grad_overflow is None or int(eng.grad_overflow) == 0


def test_engine_lazy_zero_grad_through_the_step_tape(emu, monkeypatch):
    """the recorded step keeps one backward per way the weight gradients meet the arena: store (after a lazy zero_grad) and add (accumulation
    window / eager zero_grad); replays of either equal the direct launches"""
    monkeypatch.setenv("SCOT_TAPE", "1")
    from scOT.model import ScOT
    f, meta = load_fixture("tiny_trained")
    cfg = ScOTConfig(**meta["cfg"])
    pv, t, lab, pm = fixture_inputs(meta, cfg)
    model = ScOT(cfg, compute="fp16")
    model.load_state_dict(synth_state_dict(param_shapes(cfg), meta["regime"]))
    model._ensure_arena(torch.device("cpu"))
    eng, ar = model._engine, model._arena

    def step(lazy, zero=True):
        if zero:
            model.zero_grad(lazy=lazy)
            if lazy:
                for p_ in eng._big_ptrs:
                    o = (p_ - ar.grad.data_ptr()) // 4
                    ar.grad[o:o + 64] = float("nan")
        _, _, tp = eng.forward(pv, t, lab, pm, train=True)
        model._prepare_grads()
        eng.backward(tp, torch.ones(1), None)
        return ar.grad.clone()
    ref = step(False)                                         # call 1: direct launches
    seq = [step(True), step(True), step(False), step(True), step(False)]      # call 2 records forward + the store backward; the add variant is recorded at its first use
    for g in seq:
        assert rel_l2(g.numpy(), ref.numpy()) < 1e-6
    acc = step(False, zero=False)                             # accumulate onto the last result through the recorded `add` variant
    assert rel_l2(acc.numpy(), 2.0 * ref.numpy()) < 1e-6
    ent = list(eng._taped.values())[0]
    assert ent["state"] == "ready" and set(ent["bwd"]) == {True, False}


def test_engine_window16_fast_path(emu):
    """16x16 windows, head_dim 32 (the Poseidon-B attention shape), bf16x3: the W16 kernels inside the whole program."""
    f, meta = load_fixture("tiny_w16")
    cfg = ScOTConfig(**meta["cfg"])
    pv, t, lab, pm = fixture_inputs(meta, cfg)
    model, loss, pred = run_engine(cfg, synth_state_dict(param_shapes(cfg), meta["regime"]), pv, t, lab, pm, "bf16x3")
    assert rel_l2(pred.numpy(), f["output"]) < 1e-4
    assert grads_global(model, f) < 2e-3


def test_engine_fused_block_kernels(emu, monkeypatch):
    """engine option fused_mlp (csrc/mlp_fused.hip; the default since round 2): a two-stage model with C = 96 / 192 so that all four fused
    kernels run inside the engine's forward and backward — against the layer-by-layer path and against the oracle."""
    from oracle import scot_cpu
    cfg = ScOTConfig(image_size=64, patch_size=4, num_channels=4, num_out_channels=4, embed_dim=96, depths=[1, 1], num_heads=[3, 6],
                     skip_connections=[1, 0], window_size=16, mlp_ratio=4.0, qkv_bias=True, drop_path_rate=0.0, hidden_act="gelu", p=1,
                     channel_slice_list_normalized_loss=[0, 1, 3, 4], residual_model="convnext", use_conditioning=True,
                     learn_residual=False)
    sd = synth_state_dict(param_shapes(cfg), "trained")
    pv, t, lab = synth_inputs(2, 4, 4, 64, "smooth")
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setitem(engine_mod.ENGINE_OPTIONS, "fused_mlp", flag == "1")
        model, loss, pred = run_engine(cfg, sd, pv, t, lab, None, "bf16")
        assert model._engine.fused_mlp == (flag == "1")
        res[flag] = (float(loss), pred.clone(), model._arena.grad.clone())
    with torch.no_grad():
        oloss, opred = scot_cpu.scot_forward({k: v.clone() for k, v in sd.items()}, cfg, pv, t, lab)
    e_plain, e_fused = rel_l2(res["0"][1].numpy(), opred.numpy()), rel_l2(res["1"][1].numpy(), opred.numpy())
    print(f"\\n[fused blocks] vs oracle: layer-by-layer {e_plain:.2e}, fused {e_fused:.2e}; fused vs layer-by-layer "
          f"{rel_l2(res['1'][1].numpy(), res['0'][1].numpy()):.2e}; grads {rel_l2(res['1'][2].numpy(), res['0'][2].numpy()):.2e}")
    assert e_fused < max(2e-2, 1.5 * e_plain)                       # the same bf16 error class as the validated path
    assert rel_l2(res["1"][1].numpy(), res["0"][1].numpy()) < 2e-2
    assert rel_l2(res["1"][2].numpy(), res["0"][2].numpy()) < 5e-2  # gradient arena (every parameter), two bf16 rounding realisations
    assert abs(res["1"][0] - res["0"][0]) < 5e-3 * abs(res["0"][0])


def test_engine_fused_tails_c48(emu, monkeypatch):
    """Round 6: the layer tails fused at C = 48 (Poseidon-T / -S stage 0, reference train.py:35-47; the backward in its stored-gelu' form),
    inside the engine (fp16, two blocks at C = 48 so that the next layer's qkv epilogue runs, one at C = 96): against the layer-by-layer
    forward of the same model — same loss, prediction and gradients up to the order of the fp32 sums — and against the oracle."""
    from oracle import scot_cpu
    cfg = ScOTConfig(image_size=32, patch_size=4, num_channels=4, num_out_channels=4, embed_dim=48, depths=[2, 1], num_heads=[3, 6],
                     skip_connections=[1, 0], window_size=4, mlp_ratio=4.0, qkv_bias=True, drop_path_rate=0.0, hidden_act="gelu", p=1,
                     channel_slice_list_normalized_loss=[0, 1, 3, 4], residual_model="convnext", use_conditioning=True,
                     learn_residual=False)
    sd = synth_state_dict(param_shapes(cfg), "trained")
    pv, t, lab = synth_inputs(1, 4, 4, 32, "smooth")
    res = {}
    for flag in (False, True):
        monkeypatch.setitem(engine_mod.ENGINE_OPTIONS, "fused_fwd48", flag)
        monkeypatch.setitem(engine_mod.ENGINE_OPTIONS, "fused_bwd48", flag)
        calls, bcalls = [], []
        real, realb = ops.block_tail_fwd, ops.block_tail_bwd
        monkeypatch.setattr(ops, "block_tail_fwd", lambda *a, **k: (calls.append(a[5]), real(*a, **k))[1])
        monkeypatch.setattr(ops, "block_tail_bwd", lambda *a, **k: (bcalls.append(a[7]), realb(*a, **k))[1])
        model, loss, pred = run_engine(cfg, sd, pv, t, lab, None, "fp16")
        monkeypatch.setattr(ops, "block_tail_fwd", real)
        monkeypatch.setattr(ops, "block_tail_bwd", realb)
        assert (48 in calls) == flag and (48 in bcalls) == flag, (calls, bcalls)
        res[flag] = (float(loss), pred.clone(), model._arena.grad.clone())
    with torch.no_grad():
        oloss, opred = scot_cpu.scot_forward({k: v.clone() for k, v in sd.items()}, cfg, pv, t, lab)
    e_plain, e_fused = rel_l2(res[False][1].numpy(), opred.numpy()), rel_l2(res[True][1].numpy(), opred.numpy())
    dp, dg = rel_l2(res[True][1].numpy(), res[False][1].numpy()), rel_l2(res[True][2].numpy(), res[False][2].numpy())
    print(f"\n[C = 48 forward tail] vs oracle: layer-by-layer {e_plain:.2e}, fused {e_fused:.2e}; fused vs layer-by-layer {dp:.2e}; grads {dg:.2e}")
    assert e_fused < max(4e-3, 1.5 * e_plain) and dp < 2e-3 and dg < 1e-2
    assert abs(res[True][0] - res[False][0]) < 1e-3 * abs(res[False][0])


def test_engine_lean_layer_tail(emu, monkeypatch):
    """Round 3: the ScOTLayer tail that keeps no 4C-wide tensor for the backward (forward stores neither gelu(u) nor gelu'(u), 16-bit
    pre-norm rows; backward recomputes gelu'(u), stores no du, per-workgroup partial sums instead of atomics; scot_wgrad_mlp recomputes
    gelu(u) / du) inside the engine, default fp16 mode, C = 96 / 192 — against the round-2 form of the same kernels: identical forward,
    gradients equal up to the 16-bit rounding of the pre-norm rows."""
    cfg = ScOTConfig(image_size=64, patch_size=4, num_channels=4, num_out_channels=4, embed_dim=96, depths=[2, 1], num_heads=[3, 6],
                     skip_connections=[1, 0], window_size=16, mlp_ratio=4.0, qkv_bias=True, drop_path_rate=0.0, hidden_act="gelu", p=1,
                     channel_slice_list_normalized_loss=[0, 1, 3, 4], residual_model="convnext", use_conditioning=True,
                     learn_residual=False)
    sd = synth_state_dict(param_shapes(cfg), "trained")
    pv, t, lab = synth_inputs(1, 4, 4, 64, "smooth")      # (batch 1: half the emulation time; the batch-2 indexing of the same kernels is test_engine_fused_block_kernels')
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setitem(engine_mod.ENGINE_OPTIONS, "lean_tail", flag == "1")
        from scOT.model import ScOT
        model = ScOT(cfg, compute="fp16")
        model.load_state_dict(sd)
        model._ensure_arena(torch.device("cpu"))
        eng = model._engine
        loss, pred, tape = eng.forward(pv, t, lab, None, train=True)
        recs = [r for st in tape["enc"] + tape["dec"] for r in st[0]]
        assert all(bool(r["lean"]) == (flag == "1") for r in recs) and len(recs) == 6
        assert all((r["u"] is None and r["y2"].dtype == torch.float16) == (flag == "1") for r in recs)
        model._prepare_grads()
        eng.backward(tape, torch.ones(1), None)
        res[flag] = (float(loss), pred.clone(), model._arena.grad.clone(), int(eng.grad_overflow))
    assert res["1"][0] == res["0"][0] and torch.equal(res["1"][1], res["0"][1])           # the forward does not change
    e = rel_l2(res["1"][2].numpy(), res["0"][2].numpy())
    print(f"\n[lean tail] gradient arena, lean vs round-2 form: rel-L2 {e:.2e}")
    assert e < 2e-3 and res["1"][3] == 0 and res["0"][3] == 0


@pytest.mark.parametrize("compute", ["fp16", "fp32"])
def test_engine_pooled_rows_change_nothing(emu, monkeypatch, compute):
    """Round 4: rows that are dead within a layer (the fp32 residual h, a layer's fp32 output, every inference intermediate, d_attn) come from
    engine.pool instead of fresh memory.  Same kernels on the same values: loss, prediction, gradients and the inference forward are
    BIT-identical with the pool on and off — a buffer handed out while somebody still reads it would show here (three layers per stage: both
    alternating output buffers are re-used; fp32 operands: h16 IS h and out16 IS out, which training keeps — those stay fresh)."""
    cfg = ScOTConfig(image_size=32, patch_size=4, num_channels=4, num_out_channels=4, embed_dim=96, depths=[3, 3], num_heads=[3, 6],
                     skip_connections=[1, 0], window_size=16, mlp_ratio=4.0, qkv_bias=True, drop_path_rate=0.0, hidden_act="gelu", p=1,
                     channel_slice_list_normalized_loss=[0, 1, 3, 4], residual_model="convnext", use_conditioning=True,
                     learn_residual=False)      # 64 rows per sample at C = 96 (the fused tail), 16 at C = 192 (layer-by-layer launches)
    sd = synth_state_dict(param_shapes(cfg), "trained")
    pv, t, lab = synth_inputs(1, 4, 4, 32, "smooth")
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setitem(engine_mod.ENGINE_OPTIONS, "recycle", flag == "1")
        from scOT.model import ScOT
        model = ScOT(cfg, compute=compute)
        model.load_state_dict(sd)
        model._ensure_arena(torch.device("cpu"))
        eng = model._engine
        assert eng.recycle == (flag == "1")
        loss, pred, tape = eng.forward(pv, t, lab, None, train=True)
        model._prepare_grads()
        eng.backward(tape, torch.ones(1), None)
        _, pred_eval, _ = eng.forward(pv, t, None, None, train=False)
        res[flag] = (float(loss), pred.clone(), model._arena.grad.clone(), pred_eval.clone(), len(eng._pool))
    assert res["0"][4] == 0 and res["1"][4] > 0
    assert res["1"][0] == res["0"][0] and all(torch.equal(res["1"][i], res["0"][i]) for i in (1, 2, 3))


@pytest.mark.skipif(not FULL, reason="~3 min of emulated MFMA arithmetic: SCOT_EMU_FULL=1")
@pytest.mark.parametrize("fused", ["0", "1"])
def test_engine_poseidon_T_bf16(emu, monkeypatch, fused):
    """Poseidon-T, batch 2, 128x128 (BASELINE config 2's model) forward + backward in bf16 mode, with and without the fused block
    kernels, against the real reference's fixture.  Measured here: output rel-L2 6.7e-3 / 6.9e-3 — the MI355X gives 6.7e-3."""
    monkeypatch.setitem(engine_mod.ENGINE_OPTIONS, "fused_mlp", fused == "1")
    f, meta = load_fixture("poseidonT_trained")
    cfg = ScOTConfig(**meta["cfg"])
    pv, t, lab, pm = fixture_inputs(meta, cfg)
    model, loss, pred = run_engine(cfg, synth_state_dict(param_shapes(cfg), meta["regime"]), pv, t, lab, pm, "bf16")
    assert rel_l2(pred.numpy(), f["output"]) < 2e-2
    names = [str(n) for n in f["grad_names"]]
    mine = {k: float(p.grad.double().norm()) for k, p in model.named_parameters()}
    dev = np.array([abs(mine[n] - r) / max(r, 1e-12) for n, r in zip(names, f["grad_norms"]) if r > 1e-7])
    assert np.median(dev) < 3e-2


def test_step_tape_forward_forward_backward_backward(emu, monkeypatch):
    """with the step tape on, two same-shape training forwards before their backwards (the reference's AR
    training loop, trainer.py:466-490) must not share activation buffers.  Gradients of f(a) + f(b) with the tape on == tape off,
    and the prediction returned for `a` is not overwritten by the forward of `b`."""
    from scOT.model import ScOT
    f, meta = load_fixture("tiny_trained")
    cfg = ScOTConfig(**meta["cfg"])
    sd = synth_state_dict(param_shapes(cfg), meta["regime"])
    pv, t, lab, pm = fixture_inputs(meta, cfg)
    batches = [(pv * (1.0 + 0.1 * i), t, lab + 0.05 * i) for i in range(4)]
    res = {}
    for tape_on in ("0", "1"):
        monkeypatch.setenv("SCOT_TAPE", tape_on)
        model = ScOT(cfg, compute="fp32")
        model.load_state_dict(sd)
        model._ensure_arena(torch.device("cpu"))
        eng = model._engine
        assert eng.tape_mode == (tape_on == "1")
        model._prepare_grads()
        for b in batches[:2]:                      # warm + record (forward and backward)
            _, _, tp = eng.forward(b[0], b[1], b[2], None, train=True)
            eng.backward(tp, torch.ones(1), None)
        model._arena.grad.zero_()
        la, pa, ta = eng.forward(*batches[2], None, train=True)     # replay (tape on)
        pa_copy = pa.clone()
        lb, pb, tb = eng.forward(*batches[3], None, train=True)     # must NOT reuse the recorded buffers: `ta` is pending
        assert torch.equal(pa, pa_copy), "a later forward overwrote an earlier forward's prediction"
        assert not torch.equal(pa, pb)
        eng.backward(ta, torch.ones(1), None)
        eng.backward(tb, torch.ones(1), None)
        lc, pc, tc = eng.forward(*batches[2], None, train=True)     # the tape is free again: replay
        eng.backward(tc, torch.full((1,), 0.5), None)
        res[tape_on] = (float(la), float(lb), pa.clone(), pb.clone(), model._arena.grad.clone())
    assert res["0"][0] == pytest.approx(res["1"][0], rel=1e-6) and res["0"][1] == pytest.approx(res["1"][1], rel=1e-6)
    assert rel_l2(res["1"][2].numpy(), res["0"][2].numpy()) < 1e-6 and rel_l2(res["1"][3].numpy(), res["0"][3].numpy()) < 1e-6
    assert rel_l2(res["1"][4].numpy(), res["0"][4].numpy()) < 1e-5


def test_output_attentions_match_reference(emu, monkeypatch):
    """`output_attentions=True` (tests/golden/make_attentions_fixture.py, real reference): `ScOTOutput.attentions` = the attention
    probabilities of every stage's last block, decoder stages first — recomputed from qkv and the forward's log-sum-exp by
    scot_window_attn_probs; the positional tuple of `return_dict=False` carries them behind each stack's hidden states."""
    import scOT.model as M
    monkeypatch.setattr(M, "_require_hip", lambda t: None)
    f, meta = load_fixture("tiny_attentions")
    cfg = ScOTConfig(**meta["cfg"])
    model = M.ScOT(cfg, compute="fp32")
    model.load_state_dict(synth_state_dict(param_shapes(cfg), meta["regime"]))
    model.eval()
    pv, t, lab = synth_inputs(meta["batch"], cfg.num_channels, cfg.num_out_channels, cfg.image_size, meta["kind"])
    with torch.no_grad():
        out = model(pixel_values=pv, time=t, labels=lab, output_attentions=True)
        tup = model(pixel_values=pv, time=t, labels=lab, output_attentions=True, output_hidden_states=True, return_dict=False)
        plain = model(pixel_values=pv, time=t, labels=lab)
    assert plain.attentions is None and len(out.attentions) == meta["n_attn"]
    for i, a in enumerate(out.attentions):
        ref = f[f"attn:{i}"]
        assert tuple(a.shape) == ref.shape and float((a - torch.from_numpy(ref)).abs().max()) < 3e-5, i
        assert float((a.sum(-1) - 1).abs().max()) < 1e-5
    assert rel_l2(out.output.numpy(), f["output"]) < 1e-5
    assert [(-1 if torch.is_tensor(x) else len(x)) for x in tup] == f["tuple_layout"].tolist()
    model.train()                                  # with gradients (the untaped training path), attentions are collected too
    o2 = model(pixel_values=pv, time=t, labels=lab, output_attentions=True)
    o2.loss.backward()
    assert len(o2.attentions) == 4 and float((o2.attentions[1] - out.attentions[1]).abs().max()) < 1e-6
