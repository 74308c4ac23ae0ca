"""Which parameter gradients does the fp16 mode lose?  Per-tensor relative error against the reference fixture's stored gradients
(HF-init regime: ConvNeXt layer scale 1e-6), with the gradient's magnitude relative to the largest gradient in the model."""
import sys
import numpy as np
import torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_model_gpu as T
for name in sys.argv[1:] or ["poseidonT_hf", "poseidonB_hf"]:
    f, meta = T.load_fixture(name)
    for mode in ("fp16", "bf16"):
        cfg, model = T.build(meta, mode)
        out = model(**T.inputs(cfg, meta))
        out.loss.backward()
        torch.cuda.synchronize()
        rows = []
        gmax = max(float(np.linalg.norm(f[k])) for k in f.files if k.startswith("grad:"))
        for k, p in model.named_parameters():
            key = "grad:" + k
            if key not in f.files:
                continue
            ref = f[key].astype(np.float64)
            g = p.grad.detach().cpu().numpy().astype(np.float64)
            nr = float(np.linalg.norm(ref))
            rows.append((float(np.linalg.norm(g - ref)) / max(nr, 1e-300), nr / gmax, k))
        bad = sorted([r for r in rows if r[0] > 0.05], key=lambda r: -r[1])
        print(f"[{name} {mode}] {len(rows)} stored gradient tensors, {len(bad)} with rel err > 5 %:")
        for e, m, k in bad[:12]:
            print(f"    rel err {e:8.2e}   |g| / max|g| {m:8.2e}   {k}")
