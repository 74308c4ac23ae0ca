"""Fused AdamW + global-norm gradient clipping over the model's flat parameter / gradient arenas.

The reference's training step: HF `Trainer` → `clip_grad_norm_(params, max_grad_norm)` →
`torch.optim.AdamW.step()` over the parameter groups built by `scOT/trainer.py:295-445` (restated in
`poseidon_amd.harness.optimizer_param_groups`).  With 1580 tensors (most of them 96..768-element cond-LN vectors) the stock
step is launch-bound; here parameters, gradients and both moments are single fp32 buffers, so one step is three launches
(`csrc/optim.hip`).  The class is a `torch.optim.Optimizer`: `param_groups` (and therefore LR schedulers, `zero_grad`,
HF's `create_scheduler`) work as usual; only `step()` and the state live in the arena.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional

import numpy as np
import torch

from . import lib as _lib
from . import ops
from .harness import optimizer_param_groups

SKIP = 255


def group_map8(arena, groups: List[List[str]]) -> np.ndarray:
    """uint8 map, one entry per 8 arena floats: index of the parameter group owning them, 255 where no parameter lives
    (alignment padding, the zero slot the fused qkv bias keeps for the bias-free key projection)."""
    if arena.size % 8:
        raise ValueError("arena size must be a multiple of 8 floats")
    m = np.full(arena.size // 8, SKIP, dtype=np.uint8)
    for gi, names in enumerate(groups):
        for n in names:
            o, k = arena.offsets[n], arena.numel(n)
            if o % 8:
                raise ValueError(f"{n}: arena offset {o} is not a multiple of 8 floats")
            # a tensor whose size is not a multiple of 8 shares its last chunk with alignment padding (every tensor starts on
            # a 64-float boundary): padding holds zeros with zero gradients, on which AdamW is the identity — except inside the
            # fused qkv bias, where the slot after query.bias is the (gradient-carrying, never stepped) key-bias slot
            if k % 8 and n.endswith("attention.self.query.bias"):
                raise ValueError(f"{n}: size {k} is not a multiple of 8 floats (the fused optimizer cannot separate it from the "
                                 "key-bias slot; use poseidon_amd.harness.create_optimizer for this configuration)")
            hi = (o + k + 7) // 8
            if (m[o // 8:hi] != SKIP).any():
                raise ValueError(f"{n} overlaps another parameter in the arena")
            m[o // 8:hi] = gi
    return m


class FusedAdamW(torch.optim.Optimizer):
    """AdamW (torch semantics, betas/eps as the reference: 0.9/0.999, 1e-8) over a `scOT.model.ScOT` on the GPU.

    max_grad_norm: clip the global gradient norm before the update (HF Trainer's `max_grad_norm`; None = no clipping).
    After `step()`, `last_grad_norm` is a 1-element device tensor with the pre-clip norm (what HF logs as grad_norm)."""

    def __init__(self, model, lr: float, weight_decay: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_grad_norm: Optional[float] = None, **group_kw):
        p0 = next(model.parameters(), None)
        if p0 is not None and p0.is_cuda and hasattr(model, "_ensure_arena"):
            model._ensure_arena(p0.device)      # (the flat parameter / gradient arenas are otherwise created by the first forward)
        if getattr(model, "_arena", None) is None or not model._arena.data.is_cuda:
            raise RuntimeError("FusedAdamW needs a ScOT model that lives on the GPU (call .to('cuda') first)")
        groups = optimizer_param_groups(model, weight_decay, return_names=True, **group_kw)
        names = [g.pop("names") for g in groups]
        if len(groups) > 8:
            raise ValueError("at most 8 parameter groups")
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.model = model
        arena = model._arena
        dev = arena.data.device
        self._map = torch.from_numpy(group_map8(arena, names)).to(dev)
        self.exp_avg = torch.zeros_like(arena.data)
        self.exp_avg_sq = torch.zeros_like(arena.data)
        self._nblocks = int(ops.L().scot_optim_blocks(arena.size))
        self._partial = torch.empty(self._nblocks, device=dev)
        self._clip = torch.ones(2, device=dev)
        self.max_grad_norm = max_grad_norm
        self.step_count = 0
        self._overflow_seen = torch.zeros(1, dtype=torch.int32, device=dev)   # engine.grad_overflow at the previous step

    @property
    def last_grad_norm(self) -> torch.Tensor:
        return self._clip[1:2]

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        arena = self.model._arena
        n = arena.size
        st = ops.stream()
        L = ops.L()
        clip_ptr = None
        if self.max_grad_norm is not None:
            _lib.check(L.scot_grad_sqnorm(arena.grad.data_ptr(), self._map.data_ptr(), n, self._partial.data_ptr(), st), "scot_grad_sqnorm")
            _lib.check(L.scot_clip_coef(self._partial.data_ptr(), self._nblocks, float(self.max_grad_norm), self._clip.data_ptr(), st),
                       "scot_clip_coef")
            clip_ptr = self._clip.data_ptr()
        self.step_count += 1
        g0 = self.param_groups[0]
        ng = len(self.param_groups)
        lr = (ctypes.c_float * ng)(*[float(g["lr"]) for g in self.param_groups])
        wd = (ctypes.c_float * ng)(*[float(g["weight_decay"]) for g in self.param_groups])
        b1, b2 = g0["betas"]
        # fp16 compute mode: a step whose gradients overflowed under the gradient scale is skipped ON THE DEVICE (the kernel
        # compares the engine's cumulative non-finite count with the count at the previous step), like GradScaler.step
        eng = getattr(self.model, "_engine", None)
        ovf = eng.grad_overflow if (eng is not None and eng.grad_overflow is not None) else None
        _lib.check(L.scot_adamw_step(arena.data.data_ptr(), arena.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                     self._map.data_ptr(), n, ctypes.cast(lr, ctypes.c_void_p), ctypes.cast(wd, ctypes.c_void_p), ng,
                                     float(b1), float(b2), float(g0["eps"]), self.step_count, clip_ptr,
                                     ovf.data_ptr() if ovf is not None else None,
                                     self._overflow_seen.data_ptr() if ovf is not None else None, st), "scot_adamw_step")
        if ovf is not None:
            self._overflow_seen.copy_(ovf)
        return loss

    def skipped_steps_possible(self) -> bool:
        """True when the model computes in fp16 (steps with overflowed gradients are skipped on the device)."""
        eng = getattr(self.model, "_engine", None)
        return eng is not None and eng.grad_overflow is not None

    def zero_grad(self, set_to_none: bool = False):
        self.model.zero_grad(set_to_none=False)   # one memset of the gradient arena; .grad stay views of it

    def state_dict(self) -> Dict:
        sd = super().state_dict()
        sd["fused"] = dict(step=self.step_count, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq)
        return sd

    def load_state_dict(self, sd):
        fused = sd.get("fused")
        super().load_state_dict({k: v for k, v in sd.items() if k != "fused"})
        if fused is not None:
            self.step_count = int(fused["step"])
            self.exp_avg.copy_(fused["exp_avg"])
            self.exp_avg_sq.copy_(fused["exp_avg_sq"])
