"""The two Python callers either side of the hot path, restated without HF Trainer:

  * `optimizer_param_groups` — the 4-way AdamW parameter grouping of the reference `Trainer.create_optimizer`
    (reference scOT/trainer.py:281-293, 295-397): decay = parameters outside LayerNorm-type modules whose name lacks
    "bias"; optional separate lr for embeddings/recovery (name contains "embeddings" or "patch_recovery") and for the
    time-conditioning parameters (all parameters of ConditionalLayerNorm modules).  Quirks reproduced on purpose:
    `continuous_position_bias_mlp.*.weight` has "bias" in its name → no weight decay; `logit_scale` and the ConvNeXt
    layer-scale `weight` decay; `embeddings.norm.*` go to the embeddings group, not the time-embedding group.
  * `rollout` — autoregressive evaluation/training forward of `Trainer._model_forward` (reference trainer.py:452-603):
    int n: time/n, n model calls feeding `output.detach()` back (+ pass-through of the extra input channels when
    num_channels > num_out_channels), loss averaged, or all steps stacked on dim 1 (outputs, and the hidden states /
    reshaped hidden states / attentions of every step when the model returns them); list: time = lead_time * i per step.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Union

import torch
from torch import nn


def decay_parameter_names(model: nn.Module) -> List[str]:
    from scOT.model import ConditionalLayerNorm, LayerNorm
    norm_types = (nn.LayerNorm, LayerNorm, ConditionalLayerNorm)
    inside_norm = set()
    for mname, mod in model.named_modules():
        if isinstance(mod, norm_types):
            for pname, _ in mod.named_parameters():
                inside_norm.add(f"{mname}.{pname}" if mname else pname)
    return [n for n, _ in model.named_parameters() if n not in inside_norm and "bias" not in n]


def conditional_norm_parameter_names(model: nn.Module) -> List[str]:
    from scOT.model import ConditionalLayerNorm
    out = []
    for mname, mod in model.named_modules():
        if isinstance(mod, ConditionalLayerNorm):
            out += [f"{mname}.{p}" for p, _ in mod.named_parameters()]
    return out


def optimizer_param_groups(model: nn.Module, weight_decay: float = 0.0, learning_rate_embedding_recovery: Optional[float] = None,
                           learning_rate_time_embedding: Optional[float] = None, return_names: bool = False):
    """Parameter groups in the reference's order; groups with their own lr carry an "lr" key, the others inherit the
    optimizer's default lr (exactly as the reference builds them)."""
    decay = set(decay_parameter_names(model))
    tnames = set(conditional_norm_parameter_names(model)) if learning_rate_time_embedding is not None else set()
    std, nod, emb, tim = [], [], [], []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        item = (n, p)
        if learning_rate_embedding_recovery is not None and ("embeddings" in n or "patch_recovery" in n):
            emb.append(item)
        elif n in decay:
            std.append(item)
        elif n in tnames:
            tim.append(item)
        else:
            nod.append(item)
    groups = [dict(params=std, weight_decay=weight_decay), dict(params=nod, weight_decay=0.0)]
    if learning_rate_embedding_recovery is not None:
        groups.append(dict(params=emb, lr=learning_rate_embedding_recovery, weight_decay=weight_decay))
    if learning_rate_time_embedding is not None:
        groups.append(dict(params=tim, lr=learning_rate_time_embedding, weight_decay=0.0))
    for g in groups:
        g["names"] = [n for n, _ in g["params"]]
        g["params"] = [p for _, p in g["params"]]
        if not return_names:
            g.pop("names")
    return groups


def create_optimizer(model: nn.Module, learning_rate: float, weight_decay: float = 0.0, **kw) -> torch.optim.AdamW:
    """AdamW with the reference defaults (betas 0.9/0.999, eps 1e-8; HF TrainingArguments defaults)."""
    groups = optimizer_param_groups(model, weight_decay, **kw)
    return torch.optim.AdamW(groups, lr=learning_rate, betas=(0.9, 0.999), eps=1e-8)


def compute_loss(model, inputs: Dict[str, torch.Tensor], return_outputs: bool = False, num_items_in_batch=None,
                 ar_steps: Union[int, Sequence[int], None] = None, output_all_steps: bool = False):
    """reference Trainer.compute_loss (trainer.py:605-635) without the label smoother / past-state branches no ScOT recipe uses:
    the AR forward (`rollout`), then the loss taken from the output (dict key or tuple position 0).  `num_items_in_batch` is what
    transformers >= 4.46 passes to compute_loss; like the reference (whose loss is already a mean) it is accepted and ignored."""
    outputs = rollout(model, inputs, ar_steps, output_all_steps)
    if isinstance(outputs, tuple):
        loss = outputs[0]
    else:
        if outputs.loss is None:
            raise ValueError("The model did not return a loss from the inputs, only the following keys: "
                             f"{','.join(outputs.keys())}. For reference, the inputs it received are {','.join(inputs.keys())}.")
        loss = outputs.loss
    return (loss, outputs) if return_outputs else loss


def rollout(model, inputs: Dict[str, torch.Tensor], ar_steps: Union[int, Sequence[int], None] = None,
            output_all_steps: bool = False):
    """reference Trainer._model_forward (trainer.py:452-603).  Returns the model's ScOTOutput of the last step with
    `.output` / `.loss` replaced as the reference does."""
    cfg = model.config
    if ar_steps is None or not cfg.use_conditioning:
        return model(**inputs)
    channel_difference = cfg.num_channels > cfg.num_out_channels
    if isinstance(ar_steps, int):
        schedule = [None] * ar_steps
        inputs = {**inputs, "time": inputs["time"] / ar_steps}
        lead = None
    elif isinstance(ar_steps, (list, tuple)):
        schedule = list(ar_steps)
        lead = inputs["time"]
    else:
        raise ValueError("num_ar_steps must be an integer or a list of integers.")
    outs, losses, loss = [], [], 0
    hidden, reshaped, attn = [], [], []
    outputs = None
    for i in schedule:
        if lead is not None:
            inputs = {**inputs, "time": lead * i}
        outputs = model(**inputs)
        if output_all_steps:
            outs.append(outputs.output.detach())
            if lead is not None:
                outs.append(outputs.output.detach())  # duplicated append of the reference's list branch (trainer.py:540-543)
            # per-step hidden states / attentions, stacked over the steps below (trainer.py:472-479, 509-520, 544-551, 584-595)
            if getattr(outputs, "hidden_states", None) is not None:
                hidden.append(outputs.hidden_states)
            if getattr(outputs, "attentions", None) is not None:
                attn.append(outputs.attentions)
            if getattr(outputs, "reshaped_hidden_states", None) is not None:
                reshaped.append(outputs.reshaped_hidden_states)
            if outputs.loss is not None:
                losses.append(outputs.loss)
        elif outputs.loss is not None:
            loss = loss + outputs.loss
        nxt = outputs.output.detach()
        if channel_difference:
            nxt = torch.cat([nxt, inputs["pixel_values"][:, cfg.num_out_channels:]], dim=1)
        inputs = {**inputs, "pixel_values": nxt}
    if output_all_steps:
        outputs.output = torch.stack(outs, dim=1)
        if losses:  # reference: dim 0 (int mode), dim 1 (list mode; only valid for non-scalar losses)
            outputs.loss = torch.stack(losses, dim=1 if (lead is not None and losses[0].dim() > 0) else 0)
        if hidden:
            outputs.hidden_states = [torch.stack(hs, dim=1) for hs in zip(*hidden)]
        if attn:
            outputs.attentions = [torch.stack(a, dim=1) for a in zip(*attn)]
        if reshaped:
            outputs.reshaped_hidden_states = [torch.stack(r, dim=1) for r in zip(*reshaped)]
    else:
        outputs.loss = loss / len(schedule)
    return outputs
