"""Flat parameter / gradient arenas.

All parameters of a ScOT model live in ONE contiguous fp32 buffer (and their gradients in a second one of the same
layout); the nn.Parameters of the module tree are views into it.  Why: the model has 844 (T) /
1580 (B, L) parameter tensors, most of them tiny cond-LN vectors — a flat arena makes the data-parallel gradient
exchange a handful of large RCCL all-reduces over contiguous ranges, lets the kernels accumulate gradients in place
(`+=` semantics of autograd) and lets q/k/v weights sit back-to-back so that QKV is one [3C, C] GEMM.

Layout = the reference's registration order with two local rearrangements per attention block:
  [query.weight | key.weight | value.weight]  contiguous  → fused [3C, C] weight
  [query.bias | <C zeros, no parameter> | value.bias] contiguous → fused [3C] bias (key has no bias, HF:385)
Every segment starts on a 256-byte boundary.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import torch

ALIGN = 64  # floats


def _align(n: int) -> int:
    return (n + ALIGN - 1) // ALIGN * ALIGN


def plan_layout(shapes: "OrderedDict[str, Tuple[int, ...]]") -> Tuple[Dict[str, int], int]:
    """name → element offset; plus total size.  Extra pseudo-entries `<prefix>.qkv_weight` / `.qkv_bias` give the
    offsets of the fused views."""
    offs: Dict[str, int] = {}
    cur = 0
    names = list(shapes.keys())
    done = set()
    for n in names:
        if n in done:
            continue
        if n.endswith("attention.self.query.weight"):
            pre = n[: -len("query.weight")]
            c2 = shapes[n][0] * shapes[n][1]
            c = shapes[n][0]
            cur = _align(cur)
            offs[pre + "qkv_weight"] = cur
            for i, part in enumerate(("query.weight", "key.weight", "value.weight")):
                offs[pre + part] = cur + i * c2
                done.add(pre + part)
            cur += 3 * c2
            if (pre + "query.bias") in shapes:
                cur = _align(cur)
                offs[pre + "qkv_bias"] = cur
                offs[pre + "query.bias"] = cur
                offs[pre + "value.bias"] = cur + 2 * c
                done.add(pre + "query.bias")
                done.add(pre + "value.bias")
                cur += 3 * c
            continue
        cur = _align(cur)
        offs[n] = cur
        numel = 1
        for s in shapes[n]:
            numel *= s
        cur += numel
        done.add(n)
    return offs, _align(cur)


class Arena:
    def __init__(self, shapes: "OrderedDict[str, Tuple[int, ...]]", device, requires_grad_arena: bool = True):
        self.shapes = shapes
        self.offsets, self.size = plan_layout(shapes)
        self.data = torch.zeros(self.size, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.size, dtype=torch.float32, device=device) if requires_grad_arena else None
        self._views: Dict[str, torch.Tensor] = {}
        self._gviews: Dict[str, torch.Tensor] = {}

    def numel(self, name: str) -> int:
        n = 1
        for s in self.shapes[name]:
            n *= s
        return n

    def view(self, name: str) -> torch.Tensor:
        v = self._views.get(name)
        if v is None:
            o = self.offsets[name]
            v = self.data[o:o + self.numel(name)].view(self.shapes[name])
            self._views[name] = v
        return v

    def gview(self, name: str) -> torch.Tensor:
        v = self._gviews.get(name)
        if v is None:
            o = self.offsets[name]
            v = self.grad[o:o + self.numel(name)].view(self.shapes[name])
            self._gviews[name] = v
        return v

    def span(self, name: str, numel: int, grad: bool = False) -> torch.Tensor:
        """Flat view of `numel` floats starting at entry `name` (used for the fused qkv weight / bias)."""
        o = self.offsets[name]
        return (self.grad if grad else self.data)[o:o + numel]
