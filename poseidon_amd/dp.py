"""Data-parallel gradient exchange over the flat gradient arena (RCCL over xGMI; `nccl` backend == RCCL on ROCm).

The reference delegates this to torch DDP through HF Trainer / accelerate: per step ONE mean
all-reduce of all gradients.  Here the gradients already live in one contiguous fp32 arena (poseidon_amd/arena.py), so
the exchange is a few large collectives over contiguous ranges — no per-tensor hooks, no bucket copies:

  * `GradAllReducer.allreduce()` — chunked mean all-reduce of the whole arena (optionally bf16 on the wire with fp32
    accumulation on each rank before/after), used after a graph-replayed step;
  * `GradAllReducer.ranges_in_backward_order()` + `reduce_range()` — the arena ranges in the order in which the
    backward finalises them (recovery → decoder shallow..deep → skip blocks → encoder deep..shallow → embeddings), so a
    caller can launch each range's collective on a side stream as soon as it is final (overlap with the rest of backward).
Each rank normalises its own relative loss (reference semantics: DDP-mean of per-rank losses), so the exchange is a
plain mean of gradients.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


def backward_order_groups(cfg) -> List[str]:
    nl = len(cfg.depths)
    g = ["patch_recovery."]
    g += [f"decoder.layers.{k}." for k in reversed(range(nl))]
    g += ["residual_blocks."]
    g += [f"encoder.layers.{s}." for s in reversed(range(nl))]
    g += ["embeddings."]
    return g


def group_ranges(arena, groups: List[str]) -> List[Tuple[str, int, int]]:
    """Contiguous [start, end) element ranges of the arena covering each name-prefix group."""
    out = []
    for pre in groups:
        offs = [(arena.offsets[n], arena.offsets[n] + arena.numel(n)) for n in arena.shapes if n.startswith(pre)]
        if not offs:
            continue
        out.append((pre, min(o[0] for o in offs), max(o[1] for o in offs)))
    return out


class GradAllReducer:
    def __init__(self, model, dist, wire: str = "fp32", chunk_mb: int = 128, group=None):
        self.model = model
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist is not None else 1
        self.wire = wire
        self.chunk = chunk_mb * (1 << 20) // 4
        self._wire_buf: Optional[torch.Tensor] = None
        self.comm_stream = None

    def _flat(self) -> torch.Tensor:
        return self.model.flat_grads()

    def broadcast_parameters(self, src: int = 0):
        self.dist.broadcast(self.model.flat_parameters(), src=src, group=self.group)

    def ranges_in_backward_order(self):
        return group_ranges(self.model._arena, backward_order_groups(self.model.config))

    def reduce_range(self, start: int, end: int):
        """Mean all-reduce of arena[start:end] on the current stream."""
        flat = self._flat()
        for s in range(start, end, self.chunk):
            e = min(end, s + self.chunk)
            seg = flat[s:e]
            if self.wire == "bf16":
                if self._wire_buf is None or self._wire_buf.numel() < self.chunk:
                    self._wire_buf = torch.empty(self.chunk, dtype=torch.bfloat16, device=flat.device)
                w = self._wire_buf[: e - s]
                w.copy_(seg)                                   # fp32 -> bf16 (one pass)
                self.dist.all_reduce(w, group=self.group)      # sum on the wire
                torch.mul(w, 1.0 / self.world, out=seg)        # bf16 -> fp32 and the mean's 1/N in the same pass
            else:
                seg.div_(self.world)
                self.dist.all_reduce(seg, group=self.group)

    def allreduce(self):
        flat = self._flat()
        self.reduce_range(0, flat.numel())


class OverlappedGradAllReducer(GradAllReducer):
    """Launches each arena range's mean all-reduce on a side HIP stream as soon as the backward has finalised it
    (engine.on_grads_final), so that only the last range (the small C=96 encoder stage + embeddings) is exposed.
    Roughly 73 % of Poseidon-B's gradient bytes (the two C=768 stages) are final by mid-backward."""

    def __init__(self, model, dist, wire: str = "fp32", chunk_mb: int = 64, group=None):
        super().__init__(model, dist, wire=wire, chunk_mb=chunk_mb, group=group)
        self._ranges = None
        self.comm_stream = torch.cuda.Stream()
        self._pending = False
        self._hooked = False

    def attach(self):
        self.model._engine.on_grads_final = self._on_final
        self.model._engine.reset_tapes()   # a recorded step bakes the hook in
        if not self._hooked:
            self.model.register_grad_ready_hook(lambda m: self.finish())
            self._hooked = True

    def detach(self):
        self.finish()
        self.model._engine.on_grads_final = None
        self.model._engine.reset_tapes()

    def _on_final(self, prefix: str):
        if self._ranges is None:
            self._ranges = {p: (s, e) for p, s, e in self.ranges_in_backward_order()}
        rng = self._ranges.get(prefix)
        if rng is None:
            return
        ev = torch.cuda.Event()
        ev.record()
        self.comm_stream.wait_event(ev)
        with torch.cuda.stream(self.comm_stream):
            self.reduce_range(rng[0], rng[1])
        self._pending = True

    def finish(self):
        if self._pending:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
            self._pending = False
