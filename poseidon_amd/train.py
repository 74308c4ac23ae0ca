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
(trajectories resident in HBM, a batch = one gather launch).  Callbacks, checkpoint rotation, wandb and early stopping are the HF
Trainer's own machinery and stay out of scope.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
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
                 compute_metrics: Optional[Callable[[EvalPrediction], Dict[str, Any]]] = None, optimizers: Tuple = (None, None)):
        self.model, self.args = model, args if args is not None else TrainingArguments()
        self.train_dataset, self.eval_dataset, self.compute_metrics = train_dataset, eval_dataset, compute_metrics
        self.optimizer, self.lr_scheduler = optimizers
        self.ar_steps: Union[int, Sequence[int], None] = None       # reference trainer.py:276-279
        self.output_all_steps = False
        self.label_names = ["labels"]
        self.state = dict(global_step=0, epoch=0.0, log_history=[])
        import torch.distributed as dist
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0
        self._reducer = None

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

    def train(self) -> TrainOutput:
        a, model = self.args, self.model
        if self.train_dataset is None:
            raise ValueError("Trainer: training requires a train_dataset.")
        model.train()
        n = self._num_samples(self.train_dataset)
        per_rank = math.ceil(n / self.world)
        bs, acc = a.per_device_train_batch_size, max(1, a.gradient_accumulation_steps)
        batches_per_epoch = per_rank // bs if a.dataloader_drop_last else math.ceil(per_rank / bs)
        updates_per_epoch = max(1, batches_per_epoch // acc)
        total = a.max_steps if a.max_steps > 0 else math.ceil(a.num_train_epochs * updates_per_epoch)
        epochs = math.ceil(total / updates_per_epoch)
        opt = self.create_optimizer()
        sched = self.create_scheduler(total)
        self._attach_dp()                      # (after the optimizer: it creates the arenas the exchange works on)
        overlapped = self._reducer is not None and hasattr(self._reducer, "attach")
        if overlapped:
            self._reducer.attach()             # the engine's backward announces every gradient range as it becomes final
        step, run_loss, log_loss, log_n = 0, 0.0, torch.zeros((), device=self._device()), 0
        for epoch in range(epochs):
            g = np.random.default_rng(a.seed + epoch)                 # the same permutation on every rank
            order = self._shard(g.permutation(n))
            opt.zero_grad()
            for i, batch in enumerate(self._batches(self.train_dataset, bs, order, a.dataloader_drop_last)):
                if (i // acc) >= updates_per_epoch:
                    break
                loss = self.compute_loss(model, self._model_inputs(batch))
                (loss / acc if acc > 1 else loss).backward()
                log_loss += loss.detach()
                log_n += 1
                if (i + 1) % acc:
                    continue
                if self._reducer is not None and not overlapped:
                    self._reducer.allreduce()
                opt.step()
                sched.step()
                opt.zero_grad()
                step += 1
                self.state.update(global_step=step, epoch=epoch + (i + 1) / max(1, batches_per_epoch))
                if a.logging_steps and step % a.logging_steps == 0:
                    run_loss += self._log(log_loss, log_n, sched, opt)
                    log_loss, log_n = torch.zeros_like(log_loss), 0
                if step >= total:
                    break
            if step >= total:
                break
        if log_n:
            run_loss += self._log(log_loss, log_n, sched, opt)
        if overlapped:
            self._reducer.detach()
        seen = sum(h["_n"] for h in self.state["log_history"] if "_n" in h)
        return TrainOutput(step, run_loss / max(1, seen), dict(train_steps=step, epoch=self.state["epoch"]))

    def _log(self, loss_sum: torch.Tensor, n: int, sched, opt) -> float:
        if self.dist is not None:
            self.dist.all_reduce(loss_sum)
            loss_sum = loss_sum / self.world
        v = float(loss_sum)
        entry = dict(loss=v / n, learning_rate=sched.get_last_lr()[0], step=self.state["global_step"], epoch=self.state["epoch"], _n=n)
        gn = getattr(opt, "last_grad_norm", None)
        if gn is not None and getattr(opt, "max_grad_norm", None) is not None:
            entry["grad_norm"] = float(gn)
        self.state["log_history"].append(entry)
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

    def evaluate(self, eval_dataset=None, metric_key_prefix: str = "eval") -> Dict[str, float]:
        ds = eval_dataset if eval_dataset is not None else self.eval_dataset
        if ds is None:
            raise ValueError("Trainer: evaluation requires an eval_dataset.")
        out = self._eval_loop(ds, metric_key_prefix)
        self.state["log_history"].append(dict(out.metrics, step=self.state["global_step"]))
        return out.metrics

    def predict(self, test_dataset, metric_key_prefix: str = "test") -> PredictionOutput:
        return self._eval_loop(test_dataset, metric_key_prefix)

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
