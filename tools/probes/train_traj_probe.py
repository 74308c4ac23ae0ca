import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_model_gpu as T
from scOT.trainer import FusedAdamW
for fix in ("tiny_trained", "tiny_hf"):
    f, meta = T.load_fixture(fix)
    for lr in (2e-3, 3e-4):
        res = {}
        for compute in ("fp32", "bf16x3", "fp16", "bf16"):
            cfg, model = T.build(meta, compute)
            kw = T.inputs(cfg, meta)
            model(**kw).loss.backward()
            opt = FusedAdamW(model, lr=lr, weight_decay=0.01, max_grad_norm=5.0)
            L = []
            for _ in range(12):
                opt.zero_grad(); out = model(**kw); out.loss.backward(); opt.step(); L.append(float(out.loss.detach()))
            res[compute] = np.array(L)
        a = res["fp32"]
        print(fix, "lr", lr, "fp32", np.round(a, 4).tolist())
        for k in ("bf16x3", "fp16", "bf16"):
            print("   ", k, "max rel gap %.2e" % np.max(np.abs(res[k] - a) / a), np.round(res[k], 4).tolist())
