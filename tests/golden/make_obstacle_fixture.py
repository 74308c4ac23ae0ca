"""Airfoil-style fixture from the REAL reference (run here only; /root/reference does not travel): 1→1 channels, no time
conditioning, and a (B,1,H,W) boolean `pixel_mask` (the obstacle) — the mask shape
scOT/problems/fluids/compressible.py:46-52 produces, consumed by `prediction[pixel_mask] = labels[pixel_mask]`
(scOT/model.py:1422-1423).  Forward + loss + every gradient.

usage: python tests/golden/make_obstacle_fixture.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf  # noqa: E402,F401  (installs the API-drift shim and imports the reference)
from make_fixtures import TINY, build, save  # noqa: E402
from poseidon_amd.synth import apply_obstacle, synth_inputs, synth_obstacle_mask  # noqa: E402


def main():
    torch.manual_seed(0)
    kw = dict(TINY, num_channels=1, num_out_channels=1, use_conditioning=False, channel_slice_list_normalized_loss=[0, 1])
    cfg, model = build(kw, "trained")
    batch = 3
    pv, _, lab = synth_inputs(batch, 1, 1, kw["image_size"], "smooth")
    pm = synth_obstacle_mask(batch, kw["image_size"])
    pv, lab = apply_obstacle(pv, lab, pm)
    model.zero_grad()
    out = model(pixel_values=pv, labels=lab, pixel_mask=pm)
    out.loss.backward()
    res = {"loss": out.loss.detach().numpy(), "output": out.output.detach().numpy()}
    for k, p in model.named_parameters():
        res["grad:" + k] = p.grad.detach().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
    assert np.array_equal(res["output"][pm.numpy()], lab.numpy()[pm.numpy()])
    print("masked pixels per sample", pm.flatten(1).sum(1).tolist())
    save("tiny_obstacle_mask", res, dict(cfg=kw, regime="trained", batch=batch, kind="smooth", with_mask="obstacle"))


if __name__ == "__main__":
    main()
