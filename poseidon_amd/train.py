"""Training / evaluation driver for `scOT.model.ScOT` — the caller either side of the hot path,
with the surface of the reference's `scOT/trainer.py` (an HF `Trainer` subclass) that its `train.py` / `inference.py` use:

  TrainingArguments(learning_rate, learning_rate_embedding_recovery, learning_rate_time_embedding, weight_decay, max_grad_norm,
                    num_train_epochs, per_device_*_batch_size, lr_scheduler_type, warmup_ratio, ...)     (trainer.py:235-272, train.py:277-323)
  Trainer(model, args, train_dataset, eval_dataset, compute_metrics)
      .set_ar_steps(ar_steps, output_all_steps)  ._model_forward  .compute_loss  .prediction_step        (trainer.py:447-762)
      .create_optimizer()  — the reference's four parameter groups                                       (trainer.py:281-445)
      .train()  .evaluate()  .predict(dataset, metric_key_prefix)  .save_model(dir)

What is different, on purpose: no `transformers`; the optimizer is the fused arena-wide AdamW + global-norm clip of this library
(3 launches per step, steps with overflowed fp16 gradients skipped on the device); the LR schedule is HF's linear / cosine /
constant-with-warmup as a `LambdaLR`; data parallelism is this library's gradient-arena exchange (`poseidon_amd.dp`, RCCL through
`torch.distributed`) with a DistributedSampler-style shard of every epoch's permutation — HF's DDP wrapper cannot see gradients that
live in one flat arena; datasets are either map-style datasets of sample dicts (torch DataLoader) or `DeviceTrajectories`
(trajectories resident in HBM, a batch = one gather launch).  The reference driver's own call expressions run unchanged
(train.py:277-323 `TrainingArguments(...)` with its 40 keywords, :325-328 `EarlyStoppingCallback`, :400-410 `Trainer(..., callbacks=[...])`,
`trainer.train(resume_from_checkpoint=...)`, `trainer.save_model(dir)`): per-epoch / per-N-steps evaluation and checkpoints
(`checkpoint-<step>/`: HF model directory + optimizer.pt + scheduler.pt + trainer_state.json), `save_total_limit` rotation that never
deletes the best checkpoint, `load_best_model_at_end`, resume, and the HF callback protocol (`on_train_begin / on_epoch_begin /
on_step_end / on_evaluate / on_save / on_log / on_epoch_end / on_train_end` with `TrainerState` / `TrainerControl`, so
`transformers.EarlyStoppingCallback` itself plugs in; an equivalent ships here for installations without `transformers`).  HF keywords
that configure machinery this driver does not have (wandb reporting, torch.compile, fp16 autocast, ...) are accepted and listed in ONE warning.
"""
from __future__ import annotations

import json
import math
import os
import re
import shutil
import warnings
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, NamedTuple, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .harness import compute_loss as _compute_loss
from .harness import conditional_norm_parameter_names, decay_parameter_names, optimizer_param_groups, rollout


@dataclass
class TrainingArguments:
    """The fields of HF `TrainingArguments` the reference sets (train.py:277-323) + its two extra learning rates; HF defaults."""
    output_dir: str = "./output"
    per_device_train_batch_size: int = 8
    per_device_eval_batch_size: int = 8
    num_train_epochs: float = 3.0
    max_steps: int = -1
    gradient_accumulation_steps: int = 1
    learning_rate: float = 5e-5
    learning_rate_embedding_recovery: Optional[float] = None
    learning_rate_time_embedding: Optional[float] = None
    weight_decay: float = 0.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    max_grad_norm: float = 1.0
    lr_scheduler_type: str = "linear"
    warmup_ratio: float = 0.0
    warmup_steps: int = 0
    logging_steps: int = 500
    seed: int = 42
    dataloader_num_workers: int = 0
    dataloader_drop_last: bool = False
    dp_exchange: str = "overlap"        # under torch.distributed: "overlap" (ranges all-reduced from inside the backward) | "after"
    dp_wire: str = "fp32"               # "fp32" = the reference's DDP numerics | "bf16"
    # -- evaluation / checkpoint schedule (HF semantics and defaults; reference train.py:279-315)
    overwrite_output_dir: bool = False
    evaluation_strategy: str = "no"     # "no" | "steps" | "epoch"   (HF 4.x name, the one the reference passes)
    eval_strategy: Optional[str] = None  # HF >= 4.41 name; wins when given
    eval_steps: Optional[int] = None    # default: logging_steps
    save_strategy: str = "steps"        # "no" | "steps" | "epoch"
    save_steps: int = 500
    save_total_limit: Optional[int] = None
    load_best_model_at_end: bool = False
    metric_for_best_model: Optional[str] = None
    greater_is_better: Optional[bool] = None
    logging_strategy: str = "steps"     # "no" | "steps" | "epoch"
    # -- accepted for call-site compatibility; anything but the inert value is named in one warning (see __post_init__)
    optim: str = "adamw_torch"
    eval_accumulation_steps: Optional[int] = None
    log_level: str = "passive"
    logging_nan_inf_filter: bool = True
    fp16: bool = False
    bf16: bool = False
    dataloader_pin_memory: bool = True
    gradient_checkpointing: bool = False
    auto_find_batch_size: bool = False
    full_determinism: bool = False
    torch_compile: bool = False
    report_to: Any = "none"
    run_name: Optional[str] = None
    push_to_hub: bool = False
    remove_unused_columns: bool = True
    ignored: Dict[str, Any] = field(default_factory=dict, repr=False)    # what __post_init__ found set but without effect here

    def __post_init__(self):
        if self.eval_strategy is not None:
            self.evaluation_strategy = self.eval_strategy
        for f in ("evaluation_strategy", "save_strategy", "logging_strategy"):
            v = getattr(self, f)
            v = getattr(v, "value", v)                      # an HF IntervalStrategy member
            v = "no" if v in (None, False) else str(v).lower()
            if v not in ("no", "steps", "epoch"):
                raise ValueError(f"{f}={v!r}: no | steps | epoch")
            setattr(self, f, v)
        self.eval_strategy = self.evaluation_strategy
        if self.optim not in ("adamw_torch", "adamw_hf", "adamw_torch_fused", "adamw_apex_fused"):
            raise ValueError(f"optim={self.optim!r}: this driver steps with (fused) AdamW only")
        if self.load_best_model_at_end:
            if self.metric_for_best_model is None:
                self.metric_for_best_model = "loss"         # HF default
            if self.evaluation_strategy == "no":
                raise ValueError("load_best_model_at_end requires an evaluation strategy")
            if self.save_strategy != self.evaluation_strategy:
                raise ValueError("load_best_model_at_end requires the save and eval strategy to match "
                                 f"(eval: {self.evaluation_strategy}, save: {self.save_strategy})")
        if self.metric_for_best_model is not None and self.greater_is_better is None:
            self.greater_is_better = not self.metric_for_best_model.endswith("loss")      # HF default
        inert = dict(fp16=False, bf16=False, gradient_checkpointing=False, auto_find_batch_size=False, full_determinism=False,
                     torch_compile=False, push_to_hub=False)
        ign = {k: getattr(self, k) for k, v in inert.items() if getattr(self, k) != v}
        rep = self.report_to
        if rep not in (None, "none", [], (), ["none"]):
            ign["report_to"] = rep
        self.ignored = ign
        if ign:
            warnings.warn("TrainingArguments: no effect in this driver: " + ", ".join(f"{k}={v!r}" for k, v in ign.items()) +
                          " (the compute mode is ScOT(compute=...); logs are in trainer.state.log_history)", stacklevel=3)

    def set_training(self, *, learning_rate_embedding_recovery: Optional[float] = None, learning_rate_time_embedding: Optional[float] = None,
                     **kw) -> "TrainingArguments":
        """reference trainer.py:250-260 (HF `set_training` keywords that exist here: learning_rate, batch_size -> both per-device sizes,
        weight_decay, num_epochs, max_steps, gradient_accumulation_steps, seed)"""
        m = dict(batch_size=("per_device_train_batch_size", "per_device_eval_batch_size"), num_epochs=("num_train_epochs",))
        upd: Dict[str, Any] = dict(learning_rate_embedding_recovery=learning_rate_embedding_recovery,
                                   learning_rate_time_embedding=learning_rate_time_embedding)
        for k, v in kw.items():
            for f in m.get(k, (k,)):
                if not hasattr(self, f):
                    raise TypeError(f"unknown training argument {k}")
                upd[f] = v
        for k, v in upd.items():
            setattr(self, k, v)
        return self

    def set_optimizer(self, *, learning_rate_embedding_recovery: Optional[float] = None, learning_rate_time_embedding: Optional[float] = None,
                      **kw) -> "TrainingArguments":
        """reference trainer.py:262-272 (HF `set_optimizer` keywords: learning_rate, weight_decay, beta1, beta2, epsilon)"""
        m = dict(beta1="adam_beta1", beta2="adam_beta2", epsilon="adam_epsilon")
        for k, v in kw.items():
            f = m.get(k, k)
            if k == "name":
                if v not in ("adamw_torch", "adamw_hf", "adamw_torch_fused"):
                    raise ValueError("this trainer steps with its fused AdamW only")
                continue
            if not hasattr(self, f):
                raise TypeError(f"unknown optimizer argument {k}")
            setattr(self, f, v)
        self.learning_rate_embedding_recovery = learning_rate_embedding_recovery
        self.learning_rate_time_embedding = learning_rate_time_embedding
        return self


class EvalPrediction(NamedTuple):
    predictions: np.ndarray
    label_ids: np.ndarray


class PredictionOutput(NamedTuple):
    predictions: np.ndarray
    label_ids: Optional[np.ndarray]
    metrics: Dict[str, float]


class TrainOutput(NamedTuple):
    global_step: int
    training_loss: float
    metrics: Dict[str, float]


class TrainerState(dict):
    """HF `TrainerState` fields as attributes AND as items (`trainer.state.global_step` / `trainer.state["global_step"]`)."""

    def __init__(self, **kw):
        super().__init__(global_step=0, epoch=0.0, log_history=[], best_metric=None, best_model_checkpoint=None, max_steps=0,
                         num_train_epochs=0, is_world_process_zero=True, is_local_process_zero=True, skipped_steps=0)
        self.update(kw)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v


@dataclass
class TrainerControl:
    """HF `TrainerControl`: callbacks set these, the loop reads (and resets) them."""
    should_training_stop: bool = False
    should_epoch_stop: bool = False
    should_save: bool = False
    should_evaluate: bool = False
    should_log: bool = False


class TrainerCallback:
    """HF callback protocol (every event optional): event(args, state, control, **kwargs) with kwargs model, optimizer, lr_scheduler,
    train_dataloader=None, eval_dataloader=None, and metrics (on_evaluate) / logs (on_log)."""
    EVENTS = ("on_init_end", "on_train_begin", "on_train_end", "on_epoch_begin", "on_epoch_end", "on_step_begin", "on_step_end",
              "on_evaluate", "on_predict", "on_save", "on_log")


class EarlyStoppingCallback(TrainerCallback):
    """`transformers.EarlyStoppingCallback` (the reference builds it at train.py:325-328): stop when `metric_for_best_model` has not
    improved on `state.best_metric` by more than the threshold for `early_stopping_patience` evaluations in a row."""

    def __init__(self, early_stopping_patience: int = 1, early_stopping_threshold: Optional[float] = 0.0):
        self.early_stopping_patience = early_stopping_patience
        self.early_stopping_threshold = early_stopping_threshold or 0.0
        self.early_stopping_patience_counter = 0

    def on_train_begin(self, args, state, control, **kw):
        if args.metric_for_best_model is None:
            raise AssertionError("EarlyStoppingCallback requires metric_for_best_model to be defined")
        if args.evaluation_strategy == "no":
            raise AssertionError("EarlyStoppingCallback requires IntervalStrategy of steps or epoch")

    def on_evaluate(self, args, state, control, metrics=None, **kw):
        name = args.metric_for_best_model
        name = name if name.startswith("eval_") else "eval_" + name
        v = (metrics or {}).get(name)
        if v is None:
            warnings.warn(f"early stopping needs {name}, which the evaluation did not report: disabled")
            return
        better = (v > state.best_metric if args.greater_is_better else v < state.best_metric) if state.best_metric is not None else True
        if state.best_metric is None or (better and abs(v - state.best_metric) > self.early_stopping_threshold):
            self.early_stopping_patience_counter = 0
        else:
            self.early_stopping_patience_counter += 1
        if self.early_stopping_patience_counter >= self.early_stopping_patience:
            control.should_training_stop = True


_CKPT = re.compile(r"^checkpoint-(\d+)$")


def checkpoints_in(folder: str) -> List[str]:
    """checkpoint-<step> directories under `folder`, oldest first"""
    if not os.path.isdir(folder):
        return []
    found = sorted((int(m.group(1)), d) for d in os.listdir(folder) for m in [_CKPT.match(d)] if m and os.path.isdir(os.path.join(folder, d)))
    return [os.path.join(folder, d) for _, d in found]


def lr_lambda(kind: str, warmup: int, total: int) -> Callable[[int], float]:
    """HF `get_scheduler` multipliers: "linear" (warm-up, then linear decay to 0), "cosine" (warm-up, then half a cosine to 0),
    "constant", "constant_with_warmup"."""
    def f(step: int) -> float:
        if kind != "constant" and step < warmup:
            return step / max(1, warmup)
        if kind in ("constant", "constant_with_warmup"):
            return 1.0
        prog = (step - warmup) / max(1, total - warmup)
        if kind == "linear":
            return max(0.0, 1.0 - prog)
        if kind == "cosine":
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
        raise ValueError(f"lr_scheduler_type {kind!r}: linear | cosine | constant | constant_with_warmup")
    return f


class Trainer:
    def __init__(self, model, args: Optional[TrainingArguments] = None, train_dataset=None, eval_dataset=None,
                 compute_metrics: Optional[Callable[[EvalPrediction], Dict[str, Any]]] = None, callbacks: Optional[Sequence] = None,
                 optimizers: Tuple = (None, None), data_collator=None, tokenizer=None, model_init=None,
                 preprocess_logits_for_metrics=None):
        if data_collator is not None or model_init is not None or preprocess_logits_for_metrics is not None:
            raise NotImplementedError("Trainer: data_collator / model_init / preprocess_logits_for_metrics are not supported "
                                      "(the reference passes none of them, train.py:400-407)")
        self.model, self.args = model, args if args is not None else TrainingArguments()
        self.train_dataset, self.eval_dataset, self.compute_metrics = train_dataset, eval_dataset, compute_metrics
        self.optimizer, self.lr_scheduler = optimizers
        self.ar_steps: Union[int, Sequence[int], None] = None       # reference trainer.py:276-279
        self.output_all_steps = False
        self.label_names = ["labels"]
        import torch.distributed as dist
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0
        self.state = TrainerState(is_world_process_zero=self.rank == 0, is_local_process_zero=self.rank == 0)
        self.control = TrainerControl()
        self.callbacks: List[Any] = []
        for cb in (callbacks or []):
            self.add_callback(cb)
        self._reducer = None
        self._fire("on_init_end")

    # ------------------------------------------------------------------------------------------ callbacks (HF protocol)
    def add_callback(self, callback):
        self.callbacks.append(callback() if isinstance(callback, type) else callback)

    def remove_callback(self, callback):
        self.callbacks = [c for c in self.callbacks if c is not callback and not (isinstance(callback, type) and isinstance(c, callback))]

    def _fire(self, event: str, **kw):
        for cb in self.callbacks:
            fn = getattr(cb, event, None)
            if fn is not None:
                r = fn(self.args, self.state, self.control, model=self.model, optimizer=self.optimizer, lr_scheduler=self.lr_scheduler,
                       train_dataloader=None, eval_dataloader=None, **kw)
                if r is not None:            # HF callbacks may return a new control object
                    self.control = r

    # ------------------------------------------------------------------------------------------ reference-named pieces
    def set_ar_steps(self, ar_steps=None, output_all_steps: bool = False):
        self.ar_steps, self.output_all_steps = ar_steps, output_all_steps

    def get_decay_parameter_names(self, model) -> List[str]:
        return decay_parameter_names(model)

    def get_conditional_norm_params(self, model) -> List[str]:
        return conditional_norm_parameter_names(model)

    def _model_forward(self, model, inputs):
        return rollout(model, inputs, self.ar_steps, self.output_all_steps)

    def compute_loss(self, model, inputs, return_outputs: bool = False, num_items_in_batch=None):
        return _compute_loss(model, inputs, return_outputs, num_items_in_batch, self.ar_steps, self.output_all_steps)

    def create_optimizer(self):
        """reference trainer.py:295-445: the four groups (decay / no decay / embeddings+recovery lr / time-embedding lr).  On the GPU
        the step is the fused arena-wide AdamW; a model that is not on the GPU gets torch.optim.AdamW over the same groups."""
        if self.optimizer is not None:
            return self.optimizer
        if hasattr(self.model, "_ensure_arena"):      # the flat parameter / gradient arenas (otherwise created by the first forward)
            self.model._ensure_arena(self._device())
        a = self.args
        kw = dict(learning_rate_embedding_recovery=a.learning_rate_embedding_recovery, learning_rate_time_embedding=a.learning_rate_time_embedding)
        p0 = next(self.model.parameters())
        if p0.is_cuda:
            from .optim import FusedAdamW
            self.optimizer = FusedAdamW(self.model, lr=a.learning_rate, weight_decay=a.weight_decay, betas=(a.adam_beta1, a.adam_beta2),
                                        eps=a.adam_epsilon, max_grad_norm=a.max_grad_norm if a.max_grad_norm and a.max_grad_norm > 0 else None, **kw)
        else:
            groups = optimizer_param_groups(self.model, a.weight_decay, **kw)
            self.optimizer = torch.optim.AdamW(groups, lr=a.learning_rate, betas=(a.adam_beta1, a.adam_beta2), eps=a.adam_epsilon)
        return self.optimizer

    def create_scheduler(self, num_training_steps: int):
        if self.lr_scheduler is None:
            a = self.args
            warm = a.warmup_steps if a.warmup_steps > 0 else math.ceil(num_training_steps * a.warmup_ratio)
            self.lr_scheduler = torch.optim.lr_scheduler.LambdaLR(self.create_optimizer(), lr_lambda(a.lr_scheduler_type, warm, num_training_steps))
        return self.lr_scheduler

    # ------------------------------------------------------------------------------------------ data
    def _device(self):
        return next(self.model.parameters()).device

    def _num_samples(self, ds) -> int:
        return len(ds)

    def _shard(self, order: np.ndarray) -> np.ndarray:
        """DistributedSampler semantics: pad the permutation to a multiple of the world size by wrapping, take every world-th index"""
        if self.world == 1:
            return order
        total = math.ceil(len(order) / self.world) * self.world
        order = np.concatenate([order, order[: total - len(order)]])
        return order[self.rank::self.world]

    def _batches(self, ds, batch_size: int, order: np.ndarray, drop_last: bool):
        """yields dicts of device tensors"""
        from .data import DeviceTrajectories
        dev = self._device()
        n = len(order)
        stops = range(0, n - (n % batch_size if drop_last else 0), batch_size)
        if isinstance(ds, DeviceTrajectories):
            for s in stops:
                yield ds.batch(order[s:s + batch_size])
            return
        loader = torch.utils.data.DataLoader(torch.utils.data.Subset(ds, order.tolist()), batch_size=batch_size, shuffle=False,
                                             drop_last=drop_last, num_workers=self.args.dataloader_num_workers,
                                             pin_memory=dev.type == "cuda" and self.args.dataloader_num_workers > 0)
        for b in loader:
            yield {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in b.items()}

    @staticmethod
    def _model_inputs(batch: Dict) -> Dict:
        out = dict(batch)
        if torch.is_tensor(out.get("time")):
            out["time"] = out["time"].to(torch.float32)
        return out

    # ------------------------------------------------------------------------------------------ training
    def _attach_dp(self):
        if self.dist is None or self._reducer is not None:
            return
        from .dp import GradAllReducer, OverlappedGradAllReducer
        cls = OverlappedGradAllReducer if (self.args.dp_exchange == "overlap" and self._device().type == "cuda") else GradAllReducer
        self._reducer = cls(self.model, self.dist, wire=self.args.dp_wire)
        self._reducer.broadcast_parameters(0)       # DDP's construction-time broadcast: every replica starts from rank 0's weights

    def train(self, resume_from_checkpoint: Union[bool, str, None] = None, trial=None, ignore_keys_for_eval=None, **_) -> TrainOutput:
        """reference call: `trainer.train(resume_from_checkpoint=params.resume_training)` (train.py:409).  True = the last
        `checkpoint-*` under `output_dir` (ValueError when there is none, as HF), a path = that checkpoint."""
        a, model = self.args, self.model
        if self.train_dataset is None:
            raise ValueError("Trainer: training requires a train_dataset.")
        model.train()
        n = self._num_samples(self.train_dataset)
        per_rank = math.ceil(n / self.world)
        bs, acc = a.per_device_train_batch_size, max(1, a.gradient_accumulation_steps)
        batches_per_epoch = per_rank // bs if a.dataloader_drop_last else math.ceil(per_rank / bs)
        if batches_per_epoch == 0:
            raise ValueError("Trainer: the training set holds less than one batch per rank (dataloader_drop_last=True)")
        # HF: one optimizer step per `acc` micro-batches AND at the last micro-batch of an epoch (a short tail still steps)
        updates_per_epoch = max(1, math.ceil(batches_per_epoch / acc))
        total = a.max_steps if a.max_steps > 0 else math.ceil(a.num_train_epochs * updates_per_epoch)
        epochs = math.ceil(total / updates_per_epoch)
        opt = self.create_optimizer()
        sched = self.create_scheduler(total)
        self._attach_dp()                      # (after the optimizer: it creates the arenas the exchange works on)
        # in-backward exchange only without gradient accumulation: with it, one exchange per UPDATE (after the last micro-batch)
        # instead of one per micro-batch — and attaching / detaching the hook re-records the engine's step tape
        overlapped = self._reducer is not None and hasattr(self._reducer, "attach") and acc == 1
        st = self.state
        st.update(max_steps=total, num_train_epochs=epochs)
        step, start_epoch, skip_batches = 0, 0, 0
        ckpt = None
        if resume_from_checkpoint:
            ckpt = resume_from_checkpoint if isinstance(resume_from_checkpoint, str) else (checkpoints_in(a.output_dir) or [None])[-1]
            if ckpt is None:
                raise ValueError(f"No valid checkpoint found in output directory ({a.output_dir})")
            self._load_checkpoint(ckpt, opt, sched)
            step = int(st.global_step)
            start_epoch, skip_batches = step // updates_per_epoch, (step % updates_per_epoch) * acc
        self.control = TrainerControl()
        self._fire("on_train_begin")
        if overlapped:
            self._reducer.attach()             # the engine's backward announces every gradient range as it becomes final
        run_loss, log_loss, log_n = 0.0, torch.zeros((), device=self._device()), 0
        self._skips_seen = self._skipped_steps(opt)
        for epoch in range(start_epoch, epochs):
            if step >= total or self.control.should_training_stop:
                break
            self._fire("on_epoch_begin")
            g = np.random.default_rng(a.seed + epoch)                 # the same permutation on every rank
            order = self._shard(g.permutation(n))
            opt.zero_grad()
            for i, batch in enumerate(self._batches(self.train_dataset, bs, order, a.dataloader_drop_last)):
                if epoch == start_epoch and i < skip_batches:
                    continue                                          # resumed mid-epoch: these micro-batches were consumed before the checkpoint
                last_micro = (i + 1) % acc == 0 or (i + 1) == batches_per_epoch
                if (i % acc) == 0:
                    self._fire("on_step_begin")
                loss = self.compute_loss(model, self._model_inputs(batch))
                (loss / acc if acc > 1 else loss).backward()
                log_loss += loss.detach()
                log_n += 1
                if not last_micro:
                    continue
                if self._reducer is not None and not overlapped:
                    self._reducer.allreduce()
                opt.step()
                sched.step()
                if hasattr(opt, "loss_scale_value"):      # this library's FusedAdamW
                    opt.zero_grad(overlap=True)      # the arena's fill runs beside the next forward (ScOT.zero_grad)
                else:
                    opt.zero_grad()
                step += 1
                st.update(global_step=step, epoch=epoch + (i + 1) / max(1, batches_per_epoch))
                self._fire("on_step_end")
                if (a.logging_strategy == "steps" and a.logging_steps and step % a.logging_steps == 0) or self.control.should_log:
                    run_loss += self._log(log_loss, log_n, sched, opt)
                    log_loss, log_n = torch.zeros_like(log_loss), 0
                    self.control.should_log = False
                self._maybe_evaluate_and_save("steps", step, opt, sched, ignore_keys_for_eval)
                if step >= total or self.control.should_epoch_stop or self.control.should_training_stop:
                    break
            self.control.should_epoch_stop = False
            self._fire("on_epoch_end")
            if a.logging_strategy == "epoch" and log_n:
                run_loss += self._log(log_loss, log_n, sched, opt)
                log_loss, log_n = torch.zeros_like(log_loss), 0
            self._maybe_evaluate_and_save("epoch", step, opt, sched, ignore_keys_for_eval)
        if log_n:
            run_loss += self._log(log_loss, log_n, sched, opt)
        if overlapped:
            self._reducer.detach()
        if a.load_best_model_at_end and st.best_model_checkpoint:
            self._load_weights(st.best_model_checkpoint)
        self._fire("on_train_end")
        seen = sum(h["_n"] for h in st["log_history"] if "_n" in h)
        return TrainOutput(step, run_loss / max(1, seen), dict(train_steps=step, epoch=st["epoch"]))

    # ------------------------------------------------------------------------------------------ schedule: evaluate / save / resume
    def _maybe_evaluate_and_save(self, when: str, step: int, opt, sched, ignore_keys=None):
        a, c = self.args, self.control
        metrics = None
        due_eval = (a.evaluation_strategy == when and (when == "epoch" or step % max(1, a.eval_steps or a.logging_steps) == 0)) or c.should_evaluate
        if due_eval and self.eval_dataset is not None:
            metrics = self.evaluate()
            self.model.train()
        c.should_evaluate = False
        due_save = (a.save_strategy == when and (when == "epoch" or step % max(1, a.save_steps) == 0)) or c.should_save
        if due_save:
            self._save_checkpoint(metrics, opt, sched)
        c.should_save = False

    def _skipped_steps(self, opt) -> int:
        """optimizer steps the fp16 build skipped on the device so far (overflowed gradients); a host read = a synchronisation, so
        only at logging / checkpoint points"""
        fn = getattr(opt, "skipped_steps", None)
        return int(fn()) if fn is not None else 0

    def _save_checkpoint(self, metrics: Optional[Dict[str, float]], opt, sched):
        """HF layout: <output_dir>/checkpoint-<global_step>/{config.json, model.safetensors, optimizer.pt, scheduler.pt, trainer_state.json};
        updates best_metric / best_model_checkpoint first (as HF: the early-stopping callback compares against the metric of the best SAVED model)"""
        a, st = self.args, self.state
        d = os.path.join(a.output_dir, f"checkpoint-{st.global_step}")
        if metrics is not None and a.metric_for_best_model is not None:
            name = a.metric_for_best_model if a.metric_for_best_model.startswith("eval_") else "eval_" + a.metric_for_best_model
            v = metrics.get(name)
            if v is not None and (st.best_metric is None or (v > st.best_metric if a.greater_is_better else v < st.best_metric)):
                st.best_metric, st.best_model_checkpoint = v, d
        if self.rank == 0:
            os.makedirs(d, exist_ok=True)
            self.model.save_pretrained(d)
            torch.save(opt.state_dict(), os.path.join(d, "optimizer.pt"))
            torch.save(sched.state_dict(), os.path.join(d, "scheduler.pt"))
            keep = {k: v for k, v in st.items() if k != "log_history"}
            keep["log_history"] = [{k: v for k, v in h.items()} for h in st["log_history"]]
            with open(os.path.join(d, "trainer_state.json"), "w") as f:
                json.dump(keep, f, indent=1, default=float)
            self._rotate_checkpoints()
        if self.dist is not None:
            self.dist.barrier()
        self._fire("on_save")

    def _rotate_checkpoints(self):
        a, st = self.args, self.state
        if not a.save_total_limit or a.save_total_limit <= 0:
            return
        all_ck = checkpoints_in(a.output_dir)
        limit = a.save_total_limit
        best = st.best_model_checkpoint
        if a.load_best_model_at_end and limit == 1 and best is not None and all_ck and all_ck[-1] != best:
            limit = 2                           # HF: keep the best AND the latest (needed to resume)
        victims = [c for c in all_ck if c != best]
        excess = len(all_ck) - limit
        for c in victims[:max(0, excess)]:
            shutil.rmtree(c, ignore_errors=True)

    def _load_weights(self, ckpt: str):
        st_f, bn = os.path.join(ckpt, "model.safetensors"), os.path.join(ckpt, "pytorch_model.bin")
        if os.path.exists(st_f):
            from safetensors.torch import load_file
            sd = load_file(st_f)
        elif os.path.exists(bn):
            sd = torch.load(bn, map_location="cpu", weights_only=True)
        else:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin in {ckpt}")
        self.model.load_state_dict(sd)

    def _load_checkpoint(self, ckpt: str, opt, sched):
        self._load_weights(ckpt)
        op, sp, tp = (os.path.join(ckpt, f) for f in ("optimizer.pt", "scheduler.pt", "trainer_state.json"))
        dev = self._device()
        if os.path.exists(op):
            opt.load_state_dict(torch.load(op, map_location=dev, weights_only=False))
        if os.path.exists(sp):
            sched.load_state_dict(torch.load(sp, map_location="cpu", weights_only=False))
        if os.path.exists(tp):
            with open(tp) as f:
                saved = json.load(f)
            for k in ("global_step", "epoch", "log_history", "best_metric", "best_model_checkpoint", "skipped_steps"):
                if k in saved:
                    self.state[k] = saved[k]

    def _log(self, loss_sum: torch.Tensor, n: int, sched, opt) -> float:
        if self.dist is not None:
            self.dist.all_reduce(loss_sum)
            loss_sum = loss_sum / self.world
        v = float(loss_sum)
        entry = dict(loss=v / n, learning_rate=sched.get_last_lr()[0], step=self.state["global_step"], epoch=self.state["epoch"], _n=n)
        gn = getattr(opt, "last_grad_norm", None)
        if gn is not None and getattr(opt, "max_grad_norm", None) is not None:
            entry["grad_norm"] = float(gn)
        # fp16 build: steps whose gradients overflowed were skipped on the device.  HF / GradScaler do not advance the LR schedule on
        # such a step; the count is only read here (a host read synchronises), so the schedule is rewound by the newly seen skips
        sk = self._skipped_steps(opt)
        new = sk - getattr(self, "_skips_seen", 0)
        if new > 0:
            self._skips_seen = sk
            self.state["skipped_steps"] = sk
            entry["skipped_steps"] = sk
            if hasattr(sched, "lr_lambdas") and hasattr(sched, "base_lrs"):      # the LambdaLR schedules this trainer builds
                sched.last_epoch = max(0, sched.last_epoch - new)
                for grp, lr in zip(opt.param_groups, [b * f(sched.last_epoch) for b, f in zip(sched.base_lrs, sched.lr_lambdas)]):
                    grp["lr"] = lr
                sched._last_lr = [grp["lr"] for grp in opt.param_groups]
            else:       # a scheduler passed in through `optimizers=(opt, sched)`: its state is its own, it runs `new` steps ahead
                warnings.warn(f"{type(sched).__name__} cannot be rewound: the LR schedule is {new} step(s) ahead of the applied optimizer steps")
            warnings.warn(f"{new} optimizer step(s) skipped: fp16 gradients overflowed under the loss scale "
                          f"(now {getattr(opt, 'loss_scale_value', lambda: 'n/a')()})")
        self.state["log_history"].append(entry)
        self._fire("on_log", logs=entry)
        return v

    # ------------------------------------------------------------------------------------------ evaluation
    def prediction_step(self, model, inputs: Dict, prediction_loss_only: bool, ignore_keys: Optional[List[str]] = None):
        """reference trainer.py:637-762 (the non-SageMaker branch): -> (loss, logits, labels); logits = the outputs minus "loss" and the
        ignored keys (a single remaining tensor is returned bare)."""
        ignore_keys = list(ignore_keys) if ignore_keys is not None else list(getattr(model.config, "keys_to_ignore_at_inference", []))
        has_labels = all(inputs.get(k) is not None for k in self.label_names)
        labels = tuple(inputs[k].detach() for k in self.label_names) if has_labels else None
        if labels is not None and len(labels) == 1:
            labels = labels[0]
        with torch.no_grad():
            if has_labels:
                loss, outputs = self.compute_loss(model, inputs, return_outputs=True)
                loss = loss.mean().detach()
            else:
                loss, outputs = None, self._model_forward(model, inputs)
        if prediction_loss_only:
            return loss, None, None
        if isinstance(outputs, tuple):
            logits = tuple(outputs[1:] if has_labels else outputs)
        else:
            logits = tuple(outputs[k] for k in outputs.keys() if k not in ignore_keys + ["loss"] and outputs[k] is not None)
        logits = tuple(_detach(v) for v in logits)
        return loss, (logits[0] if len(logits) == 1 else logits), labels

    def _gather(self, t: torch.Tensor) -> torch.Tensor:
        if self.dist is None:
            return t
        parts = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t.contiguous())
        return torch.stack(parts, 1).reshape(-1, *t.shape[1:])      # undo the rank-strided shard: sample k of rank r was index k·world + r

    def _eval_loop(self, ds, prefix: str) -> PredictionOutput:
        model = self.model
        was_training = model.training
        model.eval()
        n = self._num_samples(ds)
        order = self._shard(np.arange(n))
        preds, labs, losses, count = [], [], torch.zeros((), device=self._device()), 0
        for batch in self._batches(ds, self.args.per_device_eval_batch_size, order, False):
            loss, logits, labels = self.prediction_step(model, self._model_inputs(batch), prediction_loss_only=False, ignore_keys=["hidden_states", "attentions", "reshaped_hidden_states"])
            b = (logits[0] if isinstance(logits, tuple) else logits).shape[0]
            if loss is not None:
                losses += loss * b
                count += b
            preds.append(logits[0] if isinstance(logits, tuple) else logits)
            if labels is not None:
                labs.append(labels)
        P = self._gather(torch.cat(preds))[:n]
        Lb = self._gather(torch.cat(labs))[:n] if labs else None
        metrics: Dict[str, float] = {}
        if count:
            tot = torch.stack([losses, torch.tensor(float(count), device=losses.device)])
            if self.dist is not None:
                self.dist.all_reduce(tot)
            metrics[f"{prefix}_loss"] = float(tot[0] / tot[1])
        Pn, Ln = P.float().cpu().numpy(), (Lb.float().cpu().numpy() if Lb is not None else None)
        if self.compute_metrics is not None and Ln is not None:
            for k, v in self.compute_metrics(EvalPrediction(Pn, Ln)).items():
                metrics[k if k.startswith(prefix + "_") else f"{prefix}_{k}"] = v
        model.train(was_training)
        return PredictionOutput(Pn, Ln, metrics)

    def evaluate(self, eval_dataset=None, ignore_keys=None, metric_key_prefix: str = "eval") -> Dict[str, float]:
        ds = eval_dataset if eval_dataset is not None else self.eval_dataset
        if ds is None:
            raise ValueError("Trainer: evaluation requires an eval_dataset.")
        out = self._eval_loop(ds, metric_key_prefix)
        self.state["log_history"].append(dict(out.metrics, step=self.state["global_step"]))
        self._fire("on_evaluate", metrics=out.metrics)
        return out.metrics

    def predict(self, test_dataset, metric_key_prefix: str = "test", ignore_keys=None) -> PredictionOutput:
        out = self._eval_loop(test_dataset, metric_key_prefix)
        self._fire("on_predict", metrics=out.metrics)
        return out

    def save_model(self, output_dir: Optional[str] = None):
        if self.rank == 0:
            d = output_dir or self.args.output_dir
            os.makedirs(d, exist_ok=True)
            self.model.save_pretrained(d)


def _detach(v):
    if torch.is_tensor(v):
        return v.detach()
    if isinstance(v, (list, tuple)):
        return type(v)(_detach(x) for x in v)
    return v
