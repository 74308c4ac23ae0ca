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
        self._clip = torch.tensor([1.0, 0.0, 0.0], device=dev)        # {clip coefficient, gradient norm, 1 if the norm is not finite}
        self._step_state = torch.zeros(2, dtype=torch.int32, device=dev)   # {steps applied, steps skipped}: Adam's clock lives on the device
        self.max_grad_norm = max_grad_norm
        self.step_count = 0             # calls of step() (applied + skipped)
        # fp16 compute mode: torch.cuda.amp.GradScaler's schedule for the engine's gradient scale
        self.growth_factor, self.backoff_factor, self.growth_interval, self.max_scale = 2.0, 0.5, 2000, float(2 ** 30)

    @property
    def last_grad_norm(self) -> torch.Tensor:
        return self._clip[1:2]

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        arena = self.model._arena
        n = arena.size
        eng = getattr(self.model, "_engine", None)
        fp16 = eng is not None and eng.scale_state is not None
        prev = ops.use(eng.lib_kind) if eng is not None else None       # (the 16-bit copy is written in the engine's operand format)
        try:
            st = ops.stream()
            L = ops.L()
            clip_ptr = None
            if self.max_grad_norm is not None or fp16:
                # the norm of the (reduced) gradient: clipping, and — in the fp16 build — the overflow test every rank answers alike
                _lib.check(L.scot_grad_sqnorm(arena.grad.data_ptr(), self._map.data_ptr(), n, self._partial.data_ptr(), st), "scot_grad_sqnorm")
                _lib.check(L.scot_clip_coef(self._partial.data_ptr(), self._nblocks, float(self.max_grad_norm or 0.0), self._clip.data_ptr(), st),
                           "scot_clip_coef")
                clip_ptr = self._clip.data_ptr()
            self.step_count += 1
            g0 = self.param_groups[0]
            ng = len(self.param_groups)
            lr = (ctypes.c_float * ng)(*[float(g["lr"]) for g in self.param_groups])
            wd = (ctypes.c_float * ng)(*[float(g["weight_decay"]) for g in self.param_groups])
            b1, b2 = g0["betas"]
            shadow = eng.shadow if (eng is not None and eng.shadow is not None) else None
            _lib.check(L.scot_adamw_step(arena.data.data_ptr(), arena.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                         self._map.data_ptr(), n, ctypes.cast(lr, ctypes.c_void_p), ctypes.cast(wd, ctypes.c_void_p), ng,
                                         float(b1), float(b2), float(g0["eps"]), self.step_count, clip_ptr, self._step_state.data_ptr(),
                                         shadow.data_ptr() if shadow is not None else None, st), "scot_adamw_step")
            _lib.check(L.scot_optim_finish(self._step_state.data_ptr(), clip_ptr, eng.scale_state.data_ptr() if fp16 else None,
                                           float(self.growth_factor), float(self.backoff_factor),
                                           int(self.growth_interval) if fp16 else 0, float(self.max_scale), st), "scot_optim_finish")
            self.model.mark_weights_dirty()
            if shadow is not None:
                if eng.shadow_t is not None:
                    eng.transpose_weights()         # the data gradients' W^T operands, from the new master weights
                eng.weight_copies_are_current(self.model._weights_version())
        finally:
            if prev is not None:
                ops.use(prev)
        return loss

    def skipped_steps(self) -> int:
        """optimizer steps skipped on the device so far (non-finite gradient norm under the fp16 gradient scale); a host read"""
        return int(self._step_state[1])

    def applied_steps(self) -> int:
        return int(self._step_state[0])

    def loss_scale_value(self) -> float:
        eng = getattr(self.model, "_engine", None)
        return eng.grad_scale_value() if eng is not None else 1.0

    def skipped_steps_possible(self) -> bool:
        """True when the model computes in fp16 (steps with overflowed gradients are skipped on the device)."""
        eng = getattr(self.model, "_engine", None)
        return eng is not None and eng.grad_overflow is not None

    def zero_grad(self, set_to_none: bool = False, overlap: bool = False):
        self.model.zero_grad(set_to_none=False, overlap=overlap)   # one memset of the gradient arena; .grad stay views of it

    def state_dict(self) -> Dict:
        sd = super().state_dict()
        eng = getattr(self.model, "_engine", None)
        sd["fused"] = dict(step=self.step_count, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, step_state=self._step_state,
                           scale_state=eng.scale_state if (eng is not None and eng.scale_state is not None) else None)
        return sd

    def load_state_dict(self, sd):
        fused = sd.get("fused")
        super().load_state_dict({k: v for k, v in sd.items() if k != "fused"})
        if fused is not None:
            self.step_count = int(fused["step"])
            self.exp_avg.copy_(fused["exp_avg"])
            self.exp_avg_sq.copy_(fused["exp_avg_sq"])
            if fused.get("step_state") is not None:
                self._step_state.copy_(fused["step_state"])
            else:
                self._step_state.copy_(torch.tensor([self.step_count, 0], dtype=torch.int32))
            eng = getattr(self.model, "_engine", None)
            if fused.get("scale_state") is not None and eng is not None and eng.scale_state is not None:
                eng.scale_state.copy_(fused["scale_state"])
                eng._scale_ready = True
