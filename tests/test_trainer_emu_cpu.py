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


def _reference_style_arguments(TrainingArguments, ckpt_dir, config, SEED=0, CPU_CORES=0, run_name=None, resume=False):
    """The reference driver's `TrainingArguments(...)` call (reference scOT/train.py:277-323), keyword for keyword."""
    return TrainingArguments(
        output_dir=ckpt_dir,
        overwrite_output_dir=True,
        evaluation_strategy="epoch",
        per_device_train_batch_size=config["batch_size"],
        per_device_eval_batch_size=config["batch_size"],
        eval_accumulation_steps=16,
        max_grad_norm=config["max_grad_norm"],
        num_train_epochs=config["num_epochs"],
        optim="adamw_torch",
        learning_rate=config["lr"],
        learning_rate_embedding_recovery=None,
        learning_rate_time_embedding=None,
        weight_decay=config["weight_decay"],
        adam_beta1=0.9,
        adam_beta2=0.999,
        adam_epsilon=1e-8,
        lr_scheduler_type=config["lr_scheduler"],
        warmup_ratio=config["warmup_ratio"],
        log_level="passive",
        logging_strategy="steps",
        logging_steps=5,
        logging_nan_inf_filter=False,
        save_strategy="epoch",
        save_total_limit=1,
        seed=SEED,
        fp16=False,
        dataloader_num_workers=CPU_CORES,
        load_best_model_at_end=True,
        metric_for_best_model="loss",
        greater_is_better=False,
        dataloader_pin_memory=True,
        gradient_checkpointing=False,
        auto_find_batch_size=False,
        full_determinism=False,
        torch_compile=False,
        report_to="wandb",
        run_name=run_name,
    )


def test_reference_call_sites_run_unchanged(emu, tmp_path):
    """reference scOT/train.py:277-328, 400-410: TrainingArguments with all of its HF keywords, EarlyStoppingCallback,
    Trainer(..., callbacks=[...]), train(resume_from_checkpoint=...), save_model(dir) — per-epoch evaluation + checkpoint, rotation
    that keeps the best, best model restored at the end, early stopping, resume."""
    from scOT.model import ScOT
    from scOT.trainer import EarlyStoppingCallback, Trainer, TrainingArguments, checkpoints_in
    f, meta = load_fixture("tiny_trained")
    cfg = ScOTConfig(**meta["cfg"])
    sd0 = synth_state_dict(param_shapes(cfg), meta["regime"])
    train_dataset, eval_dataset = Samples(8, cfg, 0), Samples(4, cfg, 1)
    config = dict(batch_size=4, max_grad_norm=5.0, num_epochs=3, lr=2e-3, weight_decay=0.01, lr_scheduler="cosine", warmup_ratio=0.1,
                  early_stopping_patience=5)
    ckpt_dir = str(tmp_path / "run")
    with pytest.warns(UserWarning, match="report_to='wandb'"):
        train_config = _reference_style_arguments(TrainingArguments, ckpt_dir, config)
    assert train_config.ignored == {"report_to": "wandb"} and train_config.eval_strategy == "epoch"
    early_stopping = EarlyStoppingCallback(early_stopping_patience=config["early_stopping_patience"], early_stopping_threshold=0.0)
    model = ScOT(cfg, compute="fp32")
    model.load_state_dict(sd0)
    events = []

    class Spy:
        def on_epoch_end(self, args, state, control, **kw):
            events.append(("epoch_end", state.global_step))

        def on_save(self, args, state, control, **kw):
            events.append(("save", state.global_step))

        def on_evaluate(self, args, state, control, metrics=None, **kw):
            events.append(("eval", round(metrics["eval_loss"], 6)))

    def compute_metrics(eval_preds):
        return {"mean_relative_l1_error": float(np.abs(eval_preds.predictions - eval_preds.label_ids).sum() / np.abs(eval_preds.label_ids).sum())}
    trainer = Trainer(
        model=model,
        args=train_config,
        train_dataset=train_dataset,
        eval_dataset=eval_dataset,
        compute_metrics=compute_metrics,
        callbacks=[early_stopping],
    )
    trainer.add_callback(Spy())
    out = trainer.train(resume_from_checkpoint=False)
    trainer.save_model(train_config.output_dir)
    assert out.global_step == 6 and [e[0] for e in events] == ["epoch_end", "eval", "save"] * 3
    evals = [e[1] for e in events if e[0] == "eval"]
    st = trainer.state
    assert st.best_metric == pytest.approx(min(evals)) and st.best_model_checkpoint.endswith(f"checkpoint-{2 * (1 + evals.index(min(evals)))}")
    left = checkpoints_in(ckpt_dir)
    assert st.best_model_checkpoint in left and len(left) <= 2 and left[-1].endswith("checkpoint-6")    # save_total_limit=1 + best kept
    assert os.path.exists(os.path.join(left[-1], "optimizer.pt")) and os.path.exists(os.path.join(left[-1], "trainer_state.json"))
    assert os.path.exists(os.path.join(ckpt_dir, "config.json"))
    # load_best_model_at_end: the model now holds the best checkpoint's weights
    from safetensors.torch import load_file
    best = load_file(os.path.join(st.best_model_checkpoint, "model.safetensors"))
    got = model.state_dict()
    assert all(torch.equal(got[k].cpu(), v) for k, v in best.items())

    # resume: a fresh model + trainer continue from the last checkpoint (step 6 of 6 -> nothing left to do, state restored)
    model2 = ScOT(cfg, compute="fp32")
    model2.load_state_dict(sd0)
    with pytest.warns(UserWarning):
        args2 = _reference_style_arguments(TrainingArguments, ckpt_dir, dict(config, num_epochs=4))
    tr2 = Trainer(model=model2, args=args2, train_dataset=train_dataset, eval_dataset=eval_dataset, compute_metrics=compute_metrics,
                  callbacks=[EarlyStoppingCallback(early_stopping_patience=1, early_stopping_threshold=1e9)])
    out2 = tr2.train(resume_from_checkpoint=True)
    # resumed at step 6 (epoch 3 of 4); one more epoch runs; the absurd threshold makes that evaluation "no improvement": patience 1 -> stop
    assert out2.global_step == 8 and tr2.state.global_step == 8 and tr2.optimizer.state_dict() is not None
    assert any(h.get("step") == 5 for h in tr2.state["log_history"])       # the restored log history precedes the new entries
    with pytest.raises(ValueError, match="No valid checkpoint"):
        Trainer(model=model2, args=_quiet(lambda: _reference_style_arguments(TrainingArguments, str(tmp_path / "empty"), config)),
                train_dataset=train_dataset, eval_dataset=eval_dataset).train(resume_from_checkpoint=True)


def _quiet(fn):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return fn()


def test_early_stopping_stops_and_hf_callback_plugs_in(emu, tmp_path):
    from scOT.model import ScOT
    from scOT.trainer import EarlyStoppingCallback, Trainer, TrainingArguments
    f, meta = load_fixture("tiny_trained")
    cfg = ScOTConfig(**meta["cfg"])
    train_dataset, eval_dataset = Samples(4, cfg, 0), Samples(2, cfg, 1)

    def run(cb, lr):
        model = ScOT(cfg, compute="fp32")
        model.load_state_dict(synth_state_dict(param_shapes(cfg), meta["regime"]))
        args = TrainingArguments(output_dir=str(tmp_path / f"r{lr}"), evaluation_strategy="epoch", save_strategy="epoch", num_train_epochs=6,
                                 per_device_train_batch_size=2, per_device_eval_batch_size=2, learning_rate=lr, lr_scheduler_type="constant",
                                 load_best_model_at_end=True, metric_for_best_model="loss", greater_is_better=False, logging_steps=1,
                                 save_total_limit=1, max_grad_norm=0.0)
        tr = Trainer(model=model, args=args, train_dataset=train_dataset, eval_dataset=eval_dataset, callbacks=[cb])
        return tr.train(), tr
    # lr = 0: the evaluation loss never improves -> stop after `patience` evaluations without improvement (first one sets the best)
    out, tr = run(EarlyStoppingCallback(early_stopping_patience=2), 0.0)
    assert out.global_step == 2 * 3 and tr.control.should_training_stop
    try:
        from transformers import EarlyStoppingCallback as HFEarlyStopping
    except Exception:
        return
    out, tr = run(HFEarlyStopping(early_stopping_patience=2), 0.0)
    assert out.global_step == 2 * 3


def test_gradient_accumulation_steps_on_short_epoch(emu, tmp_path):
    """HF semantics: an optimizer step at every `acc`-th micro-batch AND at the epoch's last one (3 batches, acc 2 -> 2 steps per epoch)."""
    from scOT.model import ScOT
    from scOT.trainer import Trainer, TrainingArguments
    f, meta = load_fixture("tiny_trained")
    cfg = ScOTConfig(**meta["cfg"])
    model = ScOT(cfg, compute="fp32")
    model.load_state_dict(synth_state_dict(param_shapes(cfg), meta["regime"]))
    args = TrainingArguments(output_dir=str(tmp_path), per_device_train_batch_size=2, num_train_epochs=2, gradient_accumulation_steps=2,
                             learning_rate=1e-3, logging_steps=1, save_strategy="no")
    out = Trainer(model=model, args=args, train_dataset=Samples(6, cfg, 0)).train()
    assert out.global_step == 4
    args = TrainingArguments(output_dir=str(tmp_path), per_device_train_batch_size=2, num_train_epochs=1, gradient_accumulation_steps=8,
                             learning_rate=1e-3, logging_steps=1, save_strategy="no")
    assert Trainer(model=model, args=args, train_dataset=Samples(6, cfg, 0)).train().global_step == 1     # fewer batches than acc: still one step


def test_resumed_run_equals_the_uninterrupted_one(emu, tmp_path):
    """Checkpoint / resume as a property: 6 optimizer steps in one go == 4 steps, a checkpoint (model + optimizer moments + device step
    counter + scheduler + trainer state), a NEW model and trainer resumed from it, 2 more steps — same weights, same learning-rate
    trace, same data order (the epoch permutation is a function of seed and epoch; a mid-epoch resume skips the consumed batches)."""
    from scOT.model import ScOT
    from scOT.trainer import Trainer, TrainingArguments
    f, meta = load_fixture("tiny_trained")
    cfg = ScOTConfig(**meta["cfg"])
    sd = synth_state_dict(param_shapes(cfg), meta["regime"])
    train = Samples(8, cfg, 0)

    from scOT.trainer import TrainerCallback

    class StopAt(TrainerCallback):          # the "interruption": the run is laid out for 6 steps and stopped after its 4th
        def __init__(self, n):
            self.n = n

        def on_step_end(self, args, state, control, **kw):
            if state.global_step >= self.n:
                control.should_training_stop = True
            return control

    def run(out, resume=None, save_steps=4, stop_at=None):
        model = ScOT(cfg, compute="fp32")
        model.load_state_dict(sd)
        args = TrainingArguments(output_dir=str(out), per_device_train_batch_size=2, num_train_epochs=2, max_steps=6,
                                 learning_rate=2e-3, weight_decay=0.01, lr_scheduler_type="cosine", warmup_ratio=0.25, logging_steps=1,
                                 max_grad_norm=5.0, save_strategy="steps", save_steps=save_steps, seed=3)
        tr = Trainer(model=model, args=args, train_dataset=train, callbacks=[StopAt(stop_at)] if stop_at else None)
        res = tr.train(resume_from_checkpoint=resume)
        lrs = [h["learning_rate"] for h in tr.state["log_history"] if "learning_rate" in h]
        return {k: v.detach().clone() for k, v in model.state_dict().items()}, res.global_step, lrs

    straight, n0, lr0 = run(tmp_path / "a", save_steps=100)
    _, n1, _ = run(tmp_path / "b", stop_at=4)                           # step 4 = the end of epoch 1 (4 batches per epoch): checkpoint-4
    assert n0 == 6 and n1 == 4 and os.path.isdir(str(tmp_path / "b" / "checkpoint-4"))
    resumed, n2, lr2 = run(tmp_path / "b", resume=True)
    assert n2 == 6
    assert np.allclose(lr2[-2:], lr0[-2:], rtol=1e-6, atol=1e-12)
    worst = max(float((resumed[k].double() - straight[k].double()).abs().max() / (straight[k].double().abs().max() + 1e-12)) for k in straight)
    assert worst < 1e-5, worst
    # ... and interrupted in the MIDDLE of an epoch (after 2 of its 4 batches): the resumed run skips the two consumed batches
    _, n3, _ = run(tmp_path / "c", save_steps=2, stop_at=2)
    assert n3 == 2 and os.path.isdir(str(tmp_path / "c" / "checkpoint-2"))
    mid, n4, lr4 = run(tmp_path / "c", resume=True, save_steps=100)
    assert n4 == 6 and np.allclose(lr4[-4:], lr0[-4:], rtol=1e-6, atol=1e-12)
    worst = max(float((mid[k].double() - straight[k].double()).abs().max() / (straight[k].double().abs().max() + 1e-12)) for k in straight)
    assert worst < 1e-5, worst
