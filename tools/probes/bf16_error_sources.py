"""Where does the bf16 mode's output error come from?  Poseidon-T, trained-like parameters, fixture from the real reference:
inference forward with selected pieces of every ScOTLayer switched to fp32 (engine.precision_probe).
   python tools/probes/bf16_error_sources.py [poseidonT_trained|poseidonB_trained|poseidonT_hf]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_fixture, rel_l2  # noqa: E402
from test_model_gpu import build, inputs  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "poseidonT_trained"
    f, meta = load_fixture(name)
    cfg, model = build(meta, "bf16")
    kw = inputs(cfg, meta)
    model.eval()
    with torch.no_grad():
        model(**kw)
        eng = model._engine
        for ex in (None, {"attn"}, {"qkv", "attn"}, {"qkv", "attn", "proj"}, {"mlp"}, {"proj"}, {"qkv", "attn", "proj", "mlp"}):
            eng.precision_probe = ex
            out = model(**kw)
            e = rel_l2(out.output.cpu().numpy(), f["output"])
            print(f"{name}  fp32 pieces {(str(sorted(ex)) if ex else '-'):40}  output rel-L2 {e:.3e}")


if __name__ == "__main__":
    main()
