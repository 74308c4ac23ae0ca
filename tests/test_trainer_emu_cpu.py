"""poseidon_amd/train.py (the driver with the surface of the reference's scOT/trainer.py) on the CPU emulation of the kernels: argument
plumbing, the four optimizer groups, HF's LR schedules, a short training run, evaluation / prediction with the reference-style
metrics callback, AR evaluation."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_fixture
from poseidon_amd.config import ScOTConfig
from poseidon_amd.geometry import param_shapes
from poseidon_amd.synth import synth_state_dict

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "hipemu"))


@pytest.fixture
def emu(monkeypatch):
    import emu_session
    import scOT.model as M
    emu_session.patch_ops(monkeypatch, emu_session.load_emu())
    monkeypatch.setattr(M, "_require_hip", lambda t: None)     # (the product refuses CPU tensors; the emulated library takes them)
    return True


class Samples(torch.utils.data.Dataset):
    """map-style dataset of reference-style sample dicts (scOT/problems/base.py:75-86)"""

    def __init__(self, n, cfg, seed):
        g = torch.Generator().manual_seed(seed)
        R = cfg.image_size
        self.x = torch.randn(n, cfg.num_channels, R, R, generator=g)
        self.y = 0.5 * self.x[:, :cfg.num_out_channels] + 0.1 * torch.randn(n, cfg.num_out_channels, R, R, generator=g)
        self.t = torch.rand(n, generator=g)
        self.mask = torch.zeros(cfg.num_out_channels, dtype=torch.bool)

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return {"pixel_values": self.x[i], "labels": self.y[i], "time": float(self.t[i]), "pixel_mask": self.mask}


def test_lr_schedules_match_hf_formulas():
    from scOT.trainer import lr_lambda
    lin, cos, cw = lr_lambda("linear", 2, 10), lr_lambda("cosine", 2, 10), lr_lambda("constant_with_warmup", 2, 10)
    assert [lin(s) for s in (0, 1, 2, 6, 10)] == [0.0, 0.5, 1.0, 0.5, 0.0]
    assert cos(1) == 0.5 and cos(2) == 1.0 and cos(6) == pytest.approx(0.5) and cos(10) == pytest.approx(0.0, abs=1e-12)
    assert [cw(s) for s in (0, 1, 2, 9)] == [0.0, 0.5, 1.0, 1.0] and lr_lambda("constant", 2, 10)(0) == 1.0
    with pytest.raises(ValueError):
        lr_lambda("polynomial", 0, 10)(3)


def test_training_arguments_setters():
    from scOT.trainer import TrainingArguments
    a = TrainingArguments().set_training(learning_rate=1e-3, batch_size=4, num_epochs=2, learning_rate_embedding_recovery=5e-4)
    assert (a.learning_rate, a.per_device_train_batch_size, a.per_device_eval_batch_size, a.num_train_epochs) == (1e-3, 4, 4, 2)
    assert a.learning_rate_embedding_recovery == 5e-4 and a.learning_rate_time_embedding is None
    a.set_optimizer(name="adamw_torch", learning_rate=2e-3, beta2=0.95, learning_rate_time_embedding=1e-4)
    assert (a.learning_rate, a.adam_beta2, a.learning_rate_time_embedding, a.learning_rate_embedding_recovery) == (2e-3, 0.95, 1e-4, None)
    with pytest.raises(TypeError):
        a.set_training(no_such_argument=1)


def test_trainer_trains_evaluates_and_predicts(emu, tmp_path):
    from scOT.model import ScOT
    from scOT.trainer import Trainer, TrainingArguments
    f, meta = load_fixture("tiny_trained")
    cfg = ScOTConfig(**meta["cfg"])
    model = ScOT(cfg, compute="fp32")
    model.load_state_dict(synth_state_dict(param_shapes(cfg), meta["regime"]))
    train, evals = Samples(8, cfg, 0), Samples(3, cfg, 1)
    seen = {}

    def compute_metrics(p):      # the reference's callback receives numpy predictions / labels of the WHOLE evaluation set (train.py:344-398)
        seen["shapes"] = (p.predictions.shape, p.label_ids.shape)
        return {"mean_relative_l1_error": float(np.abs(p.predictions - p.label_ids).sum() / np.abs(p.label_ids).sum())}
    args = TrainingArguments(output_dir=str(tmp_path), per_device_train_batch_size=2, per_device_eval_batch_size=2, num_train_epochs=1,
                             learning_rate=2e-3, learning_rate_embedding_recovery=1e-3, learning_rate_time_embedding=5e-4, weight_decay=0.01,
                             lr_scheduler_type="cosine", warmup_ratio=0.25, logging_steps=1, max_grad_norm=5.0)
    tr = Trainer(model=model, args=args, train_dataset=train, eval_dataset=evals, compute_metrics=compute_metrics)
    before = tr.evaluate()
    assert set(before) == {"eval_loss", "eval_mean_relative_l1_error"} and seen["shapes"] == ((3, cfg.num_out_channels, 32, 32),) * 2
    out = tr.train()
    groups = tr.optimizer.param_groups
    assert len(groups) == 4 and groups[2]["initial_lr"] == 1e-3 and groups[3]["initial_lr"] == 5e-4 and groups[1]["weight_decay"] == 0.0
    hist = [h for h in tr.state["log_history"] if "learning_rate" in h]
    assert out.global_step == 4 and len(hist) == 4
    want = [2e-3 * 0.5 * (1 + math.cos(math.pi * (s - 1) / 3)) for s in (1, 2, 3, 4)]        # 1 warm-up step of 4, then half a cosine
    assert np.allclose([h["learning_rate"] for h in hist], want, rtol=1e-6, atol=1e-12)
    assert all(np.isfinite(h["loss"]) for h in hist) and out.training_loss == pytest.approx(np.mean([h["loss"] for h in hist]))
    after = tr.evaluate()
    assert after["eval_loss"] < before["eval_loss"]                  # four AdamW steps on a learnable map lower the held-out loss
    pred = tr.predict(evals, metric_key_prefix="")
    assert pred.predictions.shape == (3, cfg.num_out_channels, 32, 32) and set(pred.metrics) == {"_loss", "_mean_relative_l1_error"}
    assert pred.metrics["_loss"] == pytest.approx(after["eval_loss"], rel=1e-6)
    b = {k: (torch.stack([evals[i][k] for i in range(2)]) if torch.is_tensor(evals[0][k]) else torch.tensor([evals[i][k] for i in range(2)]))
         for k in evals[0]}
    loss, logits, labels = tr.prediction_step(model, b, prediction_loss_only=False)
    assert loss.dim() == 0 and torch.equal(labels, b["labels"]) and torch.is_tensor(logits) and logits.shape == b["labels"].shape   # (a single remaining output is returned bare, trainer.py:757-760)
    assert tr.prediction_step(model, b, prediction_loss_only=True)[1:] == (None, None)
    tr.set_ar_steps(2)                                               # AR evaluation: two model calls per sample at time / 2
    ar = tr.predict(evals, metric_key_prefix="test")
    assert ar.predictions.shape == pred.predictions.shape and np.isfinite(ar.metrics["test_loss"]) and ar.metrics["test_loss"] != after["eval_loss"]
    tr.save_model()
    assert os.path.exists(os.path.join(str(tmp_path), "config.json"))
