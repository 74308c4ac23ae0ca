"""How far are the `logit_scale` gradients of the 16-bit modes from the reference's (the one tensor family that
tests/test_model_gpu.py::test_poseidon_presets leaves out of its per-tensor bound)?   python tools/probes/logit_scale_grad.py [fixture ...]
Prints, per fixture and compute mode, every logit_scale tensor's |ref|, rel. error, and the same for the q / k projections next to it."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_model_gpu as T  # noqa: E402


def main():
    names = sys.argv[1:] or ["poseidonT_trained", "poseidonB_trained", "poseidonB_hf"]
    for name in names:
        f, meta = T.load_fixture(name)
        for compute in ("fp32", "fp16"):
            cfg, model = T.build(meta, compute)
            out = model(**T.inputs(cfg, meta))
            out.loss.backward()
            torch.cuda.synchronize()
            rows = []
            for k, p in model.named_parameters():
                if "logit_scale" not in k or "grad:" + k not in f.files:
                    continue
                ref = f["grad:" + k].astype(np.float64).ravel()
                g = p.grad.detach().cpu().numpy().astype(np.float64).ravel()
                rows.append((k, float(np.linalg.norm(ref)), float(np.linalg.norm(g - ref) / max(np.linalg.norm(ref), 1e-300)),
                             float(np.abs(g - ref).max())))
            worst = max(rows, key=lambda r: r[2])
            errs = np.array([r[2] for r in rows])
            print(f"[{name} {compute}] {len(rows)} logit_scale tensors: rel err median {np.median(errs):.2e} max {errs.max():.2e} "
                  f"(worst {worst[0]}: |ref| {worst[1]:.2e}, max abs diff {worst[3]:.2e})")
            big = [r for r in rows if r[2] > 5e-2]
            for r in big[:8]:
                print(f"    {r[0]:60s} |ref| {r[1]:.3e} rel {r[2]:.2e} abs {r[3]:.2e}")
            del model


if __name__ == "__main__":
    main()
