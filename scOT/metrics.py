"""scOT.metrics — evaluation metrics with the reference module's import path and function names
(reference scOT/metrics.py:4-55), plus the per-channel-group statistics the reference's inference driver derives from them
(reference scOT/inference.py:76-199, `compute_metrics`).

The reference works on numpy arrays after HF `Trainer.predict` has copied every prediction to the host.  These versions take
numpy arrays OR torch tensors; given GPU tensors (e.g. the `.output` of `scOT.trainer.rollout`) the reductions run on the
device and only the per-sample error vectors / the final statistics cross PCIe.  Semantics, including
the 1e-10 guard of an all-zero target and the percent scaling, are the reference's.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch


def _t(x) -> torch.Tensor:
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))


def _like(res: torch.Tensor, ref_arg):
    """numpy in → numpy out (the reference's types); tensors stay tensors (and stay on their device)."""
    return res if isinstance(ref_arg, torch.Tensor) else res.cpu().numpy()


def _sums(preds, targets, p):
    pr, tg = _t(preds), _t(targets)
    n, c = pr.shape[0], pr.shape[1]
    pr, tg = pr.reshape(n, c, -1).to(torch.float64 if pr.dtype == torch.float64 else torch.float32), tg.reshape(n, c, -1)
    tg = tg.to(pr.dtype)
    err = (pr - tg).abs().pow(p).sum(-1).sum(-1)          # Σ_channels Σ_pixels |pred - target|^p   per sample
    return err, tg


def lp_error(preds, targets, p=1):
    """(Σ_{c,x} |pred - target|^p)^(1/p) per sample → [num_samples]   (reference metrics.py:4-9)."""
    err, _ = _sums(preds, targets, p)
    return _like(err.pow(1.0 / p), preds)


def relative_lp_error(preds, targets, p=1, return_percent=True):
    """lp_error / (Σ_{c,x} |target|^p)^(1/p) per sample, in percent by default; a zero denominator is replaced by 1e-10
    (reference metrics.py:12-36)."""
    err, tg = _sums(preds, targets, p)
    norm = tg.abs().pow(p).sum(-1).sum(-1)
    norm = torch.where(norm == 0, torch.full_like(norm, 1e-10), norm)
    res = (err / norm).pow(1.0 / p)
    if return_percent:
        res = res * 100
    return _like(res, preds)


def mean_relative_lp_error(preds, targets, p=1, return_percent=True):
    e = relative_lp_error(preds, targets, p, return_percent)
    return e.mean(0) if isinstance(e, torch.Tensor) else np.mean(e, axis=0)


def median_relative_lp_error(preds, targets, p=1, return_percent=True):
    e = relative_lp_error(preds, targets, p, return_percent)
    if isinstance(e, torch.Tensor):   # numpy's median averages the two middle elements of an even-sized sample
        return torch.quantile(e.to(torch.float64), 0.5, dim=0, interpolation="midpoint").to(e.dtype)
    return np.median(e, axis=0)


def _stats(e: torch.Tensor, suffix: str) -> Dict[str, float]:
    e64 = e.to(torch.float64)
    return {
        "median_" + suffix: float(torch.quantile(e64, 0.5, interpolation="midpoint")),
        "mean_" + suffix: float(e64.mean()),
        "std_" + suffix: float(e64.std(unbiased=False)),      # np.std: population standard deviation
        "min_" + suffix: float(e64.min()),
        "max_" + suffix: float(e64.max()),
    }


def channel_group_metrics(preds, targets, channel_slice_list: Sequence[int], channel_names: Optional[Sequence[str]] = None,
                          full_data: bool = False) -> Dict[str, object]:
    """The dictionary the reference's `compute_metrics` returns (inference.py:76-199): relative and absolute L1 errors per
    channel group `[channel_slice_list[i], channel_slice_list[i+1])`, their median/mean/std/min/max over samples, and — for more
    than one group — the means over groups (`mean_relative_l1_error` = mean over the groups' means, ...).  `channel_names` are the
    dataset's `printable_channel_description`."""
    pr, tg = _t(preds), _t(targets)
    groups = [(channel_slice_list[i], channel_slice_list[i + 1]) for i in range(len(channel_slice_list) - 1)]
    rel = [_t(relative_lp_error(pr[:, a:b], tg[:, a:b], p=1, return_percent=True)) for a, b in groups]
    ab = [_t(lp_error(pr[:, a:b], tg[:, a:b], p=1)) for a, b in groups]
    rel_stats = [_stats(e, "relative_l1_error") for e in rel]
    abs_stats = [_stats(e, "l1_error") for e in ab]
    if len(groups) == 1:
        out: Dict[str, object] = {**rel_stats[0], **abs_stats[0]}
        if full_data:
            out["relative_full_data"] = rel[0].tolist()
            out["full_data"] = ab[0].tolist()
        return out
    names = list(channel_names) if channel_names is not None else [f"group{i}" for i in range(len(groups))]
    out = {
        "mean_relative_l1_error": float(np.mean([s["mean_relative_l1_error"] for s in rel_stats])),
        "mean_over_median_relative_l1_error": float(np.mean([s["median_relative_l1_error"] for s in rel_stats])),
        "mean_l1_error": float(np.mean([s["mean_l1_error"] for s in abs_stats])),
        "mean_over_median_l1_error": float(np.mean([s["median_l1_error"] for s in abs_stats])),
    }
    for i, st in enumerate(rel_stats):
        for k, v in st.items():
            out[names[i] + "/" + k] = v
        if full_data:
            out[names[i] + "/relative_full_data"] = rel[i].tolist()
    for i, st in enumerate(abs_stats):
        for k, v in st.items():
            out[names[i] + "/" + k] = v
        if full_data:
            out[names[i] + "/full_data"] = ab[i].tolist()
    return out
