"""Loss trajectories of a short fused-AdamW run in the compute modes against fp32 — and fp32 against ITSELF (atomics make two
fp32 runs differ in the last bits; AdamW's m / sqrt(v) turns that into O(lr) differences): how much of a gap is chaos."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_model_gpu as T
from scOT.trainer import FusedAdamW


def run(meta, compute, lr, steps):
    cfg, model = T.build(meta, compute)
    kw = T.inputs(cfg, meta)
    opt = FusedAdamW(model, lr=lr, weight_decay=0.01, max_grad_norm=5.0)
    L = []
    for _ in range(steps):
        opt.zero_grad(); out = model(**kw); out.loss.backward(); opt.step(); L.append(float(out.loss.detach()))
    return np.array(L)


for fix in ("tiny_trained",):
    f, meta = T.load_fixture(fix)
    for lr, steps in ((3e-4, 12), (1e-4, 12), (1e-4, 6)):
        gaps = {"fp32": [], "fp16": [], "bf16x3": []}
        for rep in range(4):
            a = run(meta, "fp32", lr, steps)
            for k in gaps:
                gaps[k].append(float(np.max(np.abs(run(meta, k, lr, steps) - a) / a)))
        print(fix, "lr", lr, "steps", steps, {k: ["%.1e" % g for g in v] for k, v in gaps.items()})
