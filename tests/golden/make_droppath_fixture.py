"""Stochastic-depth fixtures from the REAL reference (run here only; /root/reference does not travel):

  * drop_path_rates.json — the rate every ScOTLayer ends up with for Poseidon-T (rate 0.3), Poseidon-B (0.1) and the tiny
    config (0.5), read off the instantiated reference modules;
  * tiny_droppath.npz — training-mode forward + backward of the tiny model with `Swinv2DropPath` drawing its per-sample
    keep masks from a closed-form rule instead of torch.rand (the masks are stored), so the engine / the oracle can be fed
    the very same draws.

usage: python tests/golden/make_droppath_fixture.py
"""
import json
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf  # noqa: E402  (installs the API-drift shim and imports the reference)
from make_fixtures import MODEL_MAP, TINY, build, run, save  # noqa: E402

hf = mf.hf
from scOT import model as ref  # noqa: E402


def layer_rates(model):
    return {n: float(getattr(m.drop_path, "drop_prob", 0.0) or 0.0) for n, m in model.named_modules() if isinstance(m, ref.ScOTLayer)}


def main():
    torch.manual_seed(0)
    rates = {}
    base = dict(image_size=128, patch_size=4, num_channels=4, num_out_channels=4, num_heads=[3, 6, 12, 24],
                skip_connections=[2, 2, 2, 0], window_size=16, mlp_ratio=4.0, qkv_bias=True, hidden_act="gelu", p=1,
                residual_model="convnext", use_conditioning=True, learn_residual=False)
    for tag, rate in (("T", 0.3), ("B", 0.1)):
        m = ref.ScOT(ref.ScOTConfig(drop_path_rate=rate, **base, **MODEL_MAP[tag]))
        rates[tag] = dict(drop_path_rate=rate, depths=MODEL_MAP[tag]["depths"], layers=layer_rates(m))
        del m
    kw = dict(TINY, drop_path_rate=0.5)
    cfg, model = build(kw, "trained")
    rates["tiny"] = dict(drop_path_rate=0.5, depths=kw["depths"], layers=layer_rates(model))
    with open(os.path.join(HERE, "drop_path_rates.json"), "w") as f:
        json.dump(rates, f, indent=1, sort_keys=True)

    # deterministic keep masks: Swinv2DropPath.forward → drop_path(input, p, training) (HF:565-586) with the uniform draw
    # replaced by crc32(layer:branch:sample) / 2^32
    mod2name = {id(m.drop_path): n for n, m in model.named_modules() if isinstance(m, ref.ScOTLayer)}
    calls, masks = {}, {}

    def dp_forward(self, hidden_states):
        if self.drop_prob is None or self.drop_prob == 0.0 or not self.training:
            return hidden_states
        name = mod2name[id(self)]
        which = calls.get(name, 0)
        calls[name] = which + 1
        keep = 1.0 - self.drop_prob
        B = hidden_states.shape[0]
        u = torch.tensor([zlib.crc32(f"{name}:{which}:{b}".encode()) / 2.0 ** 32 for b in range(B)], dtype=hidden_states.dtype)
        rnd = torch.floor(keep + u)                       # random_tensor.floor_()
        masks[f"mask:{name}:{which}"] = (rnd / keep).numpy().astype(np.float32)
        return hidden_states.div(keep) * rnd.view(B, *([1] * (hidden_states.dim() - 1)))

    hf.Swinv2DropPath.forward = dp_forward
    model.train()
    res, _ = run(model, kw, batch=4)
    res.update(masks)
    kept = sum(int((v > 0).sum()) for v in masks.values())
    print("branches", len(masks), "kept samples", kept, "of", sum(v.size for v in masks.values()))
    save("tiny_droppath", res, dict(cfg=kw, regime="trained", batch=4, kind="smooth"))


if __name__ == "__main__":
    main()
