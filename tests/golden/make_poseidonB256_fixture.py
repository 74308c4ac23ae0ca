"""Poseidon-B at 256x256 from the REAL reference (run here only; /root/reference does not travel): BASELINE.json config 5's
shape — image_size 256 => token grids 64^2 / 32^2 / 16^2 / 8^2, SIXTEEN 16x16 windows at stage 0 and four at stage 1, both
SHIFTED in the odd blocks (reference model.py:412-440; at 128^2 stage 1 is a single unshifted window), 16x16 windows at stage 2
and 8x8 at stage 3 — batch 1, 4→4 channels, loss groups [0,1,3,4], trained-like parameters.  Stored: output, loss, the norm and
sum of every gradient and a few full gradients (same layout as poseidonT/B/L).

usage: python tests/golden/make_poseidonB256_fixture.py
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf  # noqa: E402,F401  (installs the API-drift shim and imports the reference)
from make_fixtures import build, run, save  # noqa: E402

FULL = ("embeddings.patch_embeddings.projection.weight", "patch_recovery.mixup.weight",
        "encoder.layers.0.blocks.1.attention.self.logit_scale", "encoder.layers.1.blocks.1.attention.self.logit_scale",
        "encoder.layers.1.blocks.1.attention.self.continuous_position_bias_mlp.2.weight",
        "encoder.layers.1.blocks.1.attention.self.query.weight",
        "residual_blocks.0.0.dwconv.weight", "decoder.layers.2.blocks.0.layernorm_after.weight.weight",
        "encoder.layers.0.downsample.reduction.weight")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    kw = dict(image_size=256, patch_size=4, num_channels=4, num_out_channels=4, num_heads=[3, 6, 12, 24],
              skip_connections=[2, 2, 2, 0], window_size=16, mlp_ratio=4.0, qkv_bias=True, drop_path_rate=0.0,
              hidden_act="gelu", p=1, channel_slice_list_normalized_loss=[0, 1, 3, 4], residual_model="convnext",
              use_conditioning=True, learn_residual=False, embed_dim=96, depths=[8, 8, 8, 8])
    t0 = time.time()
    cfg, model = build(kw, "trained")
    res, _ = run(model, kw, batch=1, kind="smooth")
    out = {"loss": res["loss"], "output": res["output"]}
    names = [k for k in res if k.startswith("grad:")]
    out["grad_names"] = np.array([k[5:] for k in names])
    out["grad_norms"] = np.array([float(np.linalg.norm(res[k].astype(np.float64))) for k in names])
    out["grad_sums"] = np.array([float(res[k].astype(np.float64).sum()) for k in names])
    for k in FULL:
        out["grad:" + k] = res["grad:" + k]
    save("poseidonB256_trained", out, dict(cfg=kw, regime="trained", batch=1, kind="smooth"))
    print(f"total {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
