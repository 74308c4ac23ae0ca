"""Generate golden fixtures by importing the REAL reference (build container only).

    python tests/golden/make_fixtures.py            # writes tests/golden/*.npz

The reference (/root/reference, Python) and its un-vendored dependency (`transformers`, pinned 4.29.2,
installed 5.15.0) cannot travel to the GPU box; only the small vectors written here do.  This script is
the committed provenance of those vectors.  It contains NO reference source: it imports the reference
package, applies the two-function API-drift shim described in SURVEY.md §8(c)/A.1 and records inputs/outputs.
Parameters and inputs come from poseidon_amd.synth (closed form), so fixtures store only OUTPUTS.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from poseidon_amd.synth import synth_param, synth_inputs  # noqa: E402

# ---- shim ------------------------------------------------------------------
import transformers.models.swinv2.modeling_swinv2 as hf  # noqa: E402

_orig_attn_fwd = hf.Swinv2Attention.forward


def _attn_fwd(self, hidden_states, attention_mask=None, head_mask=None, output_attentions=False):
    return _orig_attn_fwd(self, hidden_states, attention_mask, output_attentions)


hf.Swinv2Attention.forward = _attn_fwd
from scOT.model import ScOT, ScOTConfig  # noqa: E402  (the reference)

ScOT.get_head_mask = lambda self, hm, n, *a, **k: [None] * n

MODEL_MAP = {  # reference train.py:35-72 (hyper-parameter values only)
    "T": dict(embed_dim=48, depths=[4, 4, 4, 4]),
    "B": dict(embed_dim=96, depths=[8, 8, 8, 8]),
}

TINY = dict(image_size=32, patch_size=4, num_channels=4, num_out_channels=4, embed_dim=16, depths=[2, 2],
            num_heads=[1, 2], skip_connections=[1, 0], window_size=4, mlp_ratio=4.0, qkv_bias=True,
            drop_path_rate=0.0, hidden_act="gelu", p=1, channel_slice_list_normalized_loss=[0, 1, 3, 4],
            residual_model="convnext", use_conditioning=True, learn_residual=False)


def build(cfg_kw, regime):
    cfg = ScOTConfig(**cfg_kw)
    model = ScOT(cfg)
    sd = model.state_dict()
    new = {k: synth_param(k, tuple(v.shape), regime) for k, v in sd.items()}
    model.load_state_dict(new)
    model.train()  # drop_path 0 / dropout 0 → same as eval, but matches training use
    return cfg, model


def run(model, cfg_kw, batch, kind="smooth", with_mask=False, grads=True, size=None):
    size = size or cfg_kw["image_size"]
    pv, t, lab = synth_inputs(batch, cfg_kw["num_channels"], cfg_kw["num_out_channels"], size, kind)
    kw = dict(pixel_values=pv, labels=lab)
    if cfg_kw.get("use_conditioning", False):
        kw["time"] = t
    if with_mask:
        pm = torch.zeros(batch, cfg_kw["num_out_channels"], dtype=torch.bool)
        pm[:, -1] = True
        kw["pixel_mask"] = pm
    model.zero_grad()
    out = model(**kw, output_hidden_states=True)
    res = {"loss": out.loss.detach().numpy(), "output": out.output.detach().numpy()}
    if grads:
        out.loss.backward()
        for k, p in model.named_parameters():
            res["grad:" + k] = p.grad.detach().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
    return res, out


def save(name, res, meta):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, __meta__=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **res)
    print(f"{name}: {os.path.getsize(path)/1024:.0f} KB, loss={float(res['loss']) if np.ndim(res['loss'])==0 else res['loss']}")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    # 1. tiny whole-model fixtures: all grads, both regimes, + hidden states of the trained regime
    for regime in ("trained", "hf"):
        cfg, model = build(TINY, regime)
        res, out = run(model, TINY, batch=2)
        if regime == "trained":
            # encoder hidden states: (embeddings, s0, s1) ; decoder: (input, after stage0, after stage1)
            n_dec = len(TINY["depths"]) + 1
            hs = out.hidden_states
            for i, h in enumerate(hs[:n_dec]):
                res[f"dec_hidden:{i}"] = h.detach().numpy()
            for i, h in enumerate(hs[n_dec:]):
                res[f"enc_hidden:{i}"] = h.detach().numpy()
        save(f"tiny_{regime}", res, dict(cfg=TINY, regime=regime, batch=2, kind="smooth"))

    # 2. odd grid: 36/4 = 9 → layer padding (9→12, 5→8), merge padding (9→10), unmerge crop (10→9)
    odd = dict(TINY, image_size=36)
    cfg, model = build(odd, "trained")
    res, _ = run(model, odd, batch=2)
    save("tiny_odd", res, dict(cfg=odd, regime="trained", batch=2, kind="smooth"))

    # 3. shifted windows at both stages + 3 stages + head_dim 8/16 mix, window 4 on 64x64 → grids 16, 8, 4
    sh = dict(TINY, image_size=64, embed_dim=16, depths=[2, 2, 2], num_heads=[1, 2, 4], skip_connections=[2, 1, 0])
    cfg, model = build(sh, "trained")
    res, _ = run(model, sh, batch=2)
    save("tiny_shift3", res, dict(cfg=sh, regime="trained", batch=2, kind="smooth"))

    # 4. variants: no conditioning (plain LayerNorm, time None), p=2 ungrouped loss, learn_residual + pixel_mask,
    #    5→4 channels (channel difference), window 16 with head_dim 32 (Poseidon-B attention shape, N=256)
    nocond = dict(TINY, use_conditioning=False, channel_slice_list_normalized_loss=None, p=2)
    cfg, model = build(nocond, "trained")
    res, _ = run(model, nocond, batch=2)
    save("tiny_nocond_p2", res, dict(cfg=nocond, regime="trained", batch=2, kind="smooth"))

    lr = dict(TINY, learn_residual=True, num_channels=5, channel_slice_list_normalized_loss=[0, 1, 3, 4])
    cfg, model = build(lr, "trained")
    res, _ = run(model, lr, batch=2, with_mask=True)
    save("tiny_learnres_mask", res, dict(cfg=lr, regime="trained", batch=2, kind="smooth", with_mask=True))

    w16 = dict(TINY, image_size=128, embed_dim=32, depths=[2, 2], num_heads=[1, 2], window_size=16, skip_connections=[1, 0])
    cfg, model = build(w16, "trained")
    res, _ = run(model, w16, batch=1)
    keep = {k: v for k, v in res.items() if not k.startswith("grad:")}
    for k, v in res.items():  # keep grads of the attention-specific params + a few others (size)
        if k.startswith("grad:") and any(s in k for s in ("logit_scale", "continuous_position_bias", "embeddings", "patch_recovery",
                                                           "layernorm_before", "query", "key.weight")):
            keep[k] = v
    save("tiny_w16", keep, dict(cfg=w16, regime="trained", batch=1, kind="smooth"))

    # 5. spectral resize path: tiny config (image_size 32) fed 64x64 and 16x16 inputs
    cfg, model = build(TINY, "trained")
    for size in (64, 16):
        res, _ = run(model, TINY, batch=2, grads=False, size=size)
        save(f"tiny_resize{size}", res, dict(cfg=TINY, regime="trained", batch=2, kind="smooth", size=size))

    # 6. Poseidon-T / Poseidon-B @128x128x4 (BASELINE configs 1-3): output + loss + grad norms
    for tag, batch in (("T", 2), ("B", 1)):
        kw = dict(image_size=128, patch_size=4, num_channels=4, num_out_channels=4, num_heads=[3, 6, 12, 24],
                  skip_connections=[2, 2, 2, 0], window_size=16, mlp_ratio=4.0, qkv_bias=True, drop_path_rate=0.0,
                  hidden_act="gelu", p=1, channel_slice_list_normalized_loss=[0, 1, 3, 4], residual_model="convnext",
                  use_conditioning=True, learn_residual=False, **MODEL_MAP[tag])
        for regime in ("trained", "hf"):
            cfg, model = build(kw, regime)
            res, _ = run(model, kw, batch=batch, kind="noise" if regime == "hf" else "smooth")
            out = {"loss": res["loss"], "output": res["output"]}
            names = [k for k in res if k.startswith("grad:")]
            out["grad_names"] = np.array([k[5:] for k in names])
            out["grad_norms"] = np.array([float(np.linalg.norm(res[k].astype(np.float64))) for k in names])
            out["grad_sums"] = np.array([float(res[k].astype(np.float64).sum()) for k in names])
            # a few full grads that exercise every backward kernel
            for k in ("embeddings.patch_embeddings.projection.weight", "patch_recovery.mixup.weight",
                      "encoder.layers.0.blocks.1.attention.self.logit_scale",
                      "encoder.layers.0.blocks.1.attention.self.continuous_position_bias_mlp.2.weight",
                      "residual_blocks.0.0.dwconv.weight", "decoder.layers.3.blocks.0.layernorm_after.weight.weight",
                      "encoder.layers.3.blocks.0.attention.self.key.weight" if tag == "T" else
                      "encoder.layers.0.downsample.reduction.weight"):
                out["grad:" + k] = res["grad:" + k]
            save(f"poseidon{tag}_{regime}", out, dict(cfg=kw, regime=regime, batch=batch,
                                                      kind="noise" if regime == "hf" else "smooth"))
            del model

    # 7. harness pins: optimizer group membership + AR rollout
    from scOT.trainer import Trainer, TrainingArguments  # noqa: E402  (the reference)
    kwT = dict(image_size=128, patch_size=4, num_channels=4, num_out_channels=4, num_heads=[3, 6, 12, 24],
               skip_connections=[2, 2, 2, 0], window_size=16, mlp_ratio=4.0, qkv_bias=True, drop_path_rate=0.0,
               hidden_act="gelu", p=1, channel_slice_list_normalized_loss=[0, 1, 3, 4], residual_model="convnext",
               use_conditioning=True, learn_residual=False, **MODEL_MAP["T"])
    cfg, model = build(kwT, "hf")
    args = TrainingArguments(output_dir="/tmp/_fx", report_to="none", use_cpu=True, learning_rate=5e-4,
                             learning_rate_embedding_recovery=1e-4, learning_rate_time_embedding=2e-4,
                             weight_decay=1e-6)
    tr = Trainer(model=model, args=args)
    opt = tr.create_optimizer()
    id2name = {id(p): n for n, p in model.named_parameters()}
    groups = {}
    for gi, g in enumerate(opt.param_groups):
        groups[f"group{gi}"] = dict(lr=g["lr"], weight_decay=g["weight_decay"],
                                    names=sorted(id2name[id(p)] for p in g["params"]))
    with open(os.path.join(HERE, "optimizer_groups_T.json"), "w") as f:
        json.dump(groups, f)
    print({k: (len(v["names"]), v["lr"], v["weight_decay"]) for k, v in groups.items()})

    cfg, model = build(TINY, "trained")
    model.eval()
    tr = Trainer(model=model, args=TrainingArguments(output_dir="/tmp/_fx", report_to="none", use_cpu=True))
    pv, t, lab = synth_inputs(2, 4, 4, 32, "smooth")
    roll = {}
    with torch.no_grad():
        tr.set_ar_steps(3)
        o = tr._model_forward(model, dict(pixel_values=pv, time=t, labels=lab))
        roll["int3_output"], roll["int3_loss"] = o.output.numpy(), o.loss.numpy()
        tr.ar_steps, tr.output_all_steps = None, False
        tr.set_ar_steps(2, output_all_steps=True)
        o = tr._model_forward(model, dict(pixel_values=pv, time=t, labels=lab))
        roll["int2all_output"], roll["int2all_loss"] = o.output.numpy(), o.loss.numpy()
        tr.ar_steps, tr.output_all_steps = None, False
        tr.set_ar_steps([1, 2])
        o = tr._model_forward(model, dict(pixel_values=pv, time=t * 0.5, labels=lab))
        roll["list12_output"], roll["list12_loss"] = o.output.numpy(), o.loss.numpy()
    np.savez_compressed(os.path.join(HERE, "rollout_tiny.npz"), **roll)
    print("rollout:", {k: v.shape for k, v in roll.items()})


if __name__ == "__main__":
    main()
