"""Pins for scOT/metrics.py from the REAL reference's metrics module (run here only: /root/reference does not travel).
   python tests/golden/make_metrics_pins.py  →  tests/golden/metrics_pins.json"""
import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from poseidon_amd.synth import closed_form_tensor  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_metrics", "/root/reference/scOT/metrics.py")
rm = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rm)


def inputs():
    pr = np.asarray(closed_form_tensor("metrics:pred", (6, 4, 16, 16), 1.0), dtype=np.float32)
    tg = np.asarray(closed_form_tensor("metrics:target", (6, 4, 16, 16), 1.0), dtype=np.float32)
    tg[3] = 0.0   # an all-zero target exercises the 1e-10 guard
    return pr, tg


def main():
    pr, tg = inputs()
    out = {}
    for p in (1, 2):
        out[f"lp_error_p{p}"] = rm.lp_error(pr, tg, p).tolist()
        out[f"relative_lp_error_p{p}"] = rm.relative_lp_error(pr, tg, p).tolist()
        out[f"relative_lp_error_p{p}_nopercent"] = rm.relative_lp_error(pr, tg, p, return_percent=False).tolist()
        out[f"mean_relative_p{p}"] = float(rm.mean_relative_lp_error(pr, tg, p))
        out[f"median_relative_p{p}"] = float(rm.median_relative_lp_error(pr, tg, p))
    groups = [0, 1, 3, 4]
    for i in range(3):
        a, b = groups[i], groups[i + 1]
        e = rm.relative_lp_error(pr[:, a:b], tg[:, a:b], p=1, return_percent=True)
        l1 = rm.lp_error(pr[:, a:b], tg[:, a:b], p=1)
        out[f"group{i}"] = dict(median_rel=float(np.median(e)), mean_rel=float(np.mean(e)), std_rel=float(np.std(e)),
                                min_rel=float(np.min(e)), max_rel=float(np.max(e)), median_abs=float(np.median(l1)),
                                mean_abs=float(np.mean(l1)), std_abs=float(np.std(l1)))
    json.dump(out, open(os.path.join(HERE, "metrics_pins.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
