"""`ScOT(config, use_mask_token=True)` + `bool_masked_pos` from the REAL reference (run here only): masked patch positions take the
learned mask token after the embedding norm (reference model.py:323-327, 353-359).  Tiny config, trained-like parameters, batch 2;
the mask is a closed-form pattern (position (b, l) masked iff (3 l + 5 b) % 7 < 2); output, loss and all gradients.

usage: python tests/golden/make_masktoken_fixture.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf  # noqa: E402,F401  (installs the API-drift shim and imports the reference)
from make_fixtures import TINY, ScOT, ScOTConfig, save, synth_inputs, synth_param  # noqa: E402


def mask_pattern(batch, npatch):
    b = torch.arange(batch).view(-1, 1)
    l = torch.arange(npatch).view(1, -1)
    return ((3 * l + 5 * b) % 7) < 2


def main():
    torch.manual_seed(0)
    cfg = ScOTConfig(**TINY)
    model = ScOT(cfg, use_mask_token=True)
    model.load_state_dict({k: synth_param(k, tuple(v.shape), "trained") for k, v in model.state_dict().items()})
    model.train()
    pv, t, lab = synth_inputs(2, 4, 4, 32, "smooth")
    mask = mask_pattern(2, (32 // 4) ** 2)
    out = model(pixel_values=pv, time=t, labels=lab, bool_masked_pos=mask)
    out.loss.backward()
    res = {"loss": out.loss.detach().numpy(), "output": out.output.detach().numpy()}
    for k, p in model.named_parameters():
        res["grad:" + k] = p.grad.detach().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
    assert float(np.abs(res["grad:embeddings.mask_token"]).sum()) > 0
    save("tiny_masktoken", res, dict(cfg=TINY, regime="trained", batch=2, kind="smooth", mask_token=True))


if __name__ == "__main__":
    main()
