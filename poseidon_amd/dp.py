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


SPLIT_FRACTION = 8      # a stage holding more than 1/8 of the parameters is announced in two halves (see stage_split)


def stage_split(cfg, stage_prefix: str) -> int:
    """Block index at which stage `stage_prefix` ("encoder.layers.3." / "decoder.layers.0.") is announced in two halves, 0 = whole.
    The widest stages (Poseidon-B: 8 blocks at C = 768 in the encoder's last and the decoder's first stage, 36 % of the gradient bytes
    each, reference train.py:35-72) finish half of their blocks' gradients — 114 MB — four layers before the other half: announcing
    `blocks[s:]` (+ the stage's resampling layer, whose backward runs first) when the backward has passed block s lets the first
    collective start after 4 layers of backward instead of 8.  Pure function of the configuration: engine and exchange agree by
    construction."""
    side, idx = stage_prefix.split(".")[0], int(stage_prefix.split(".")[2])
    nl = len(cfg.depths)
    depth = int(cfg.depths[idx if side == "encoder" else nl - 1 - idx])
    width = int(cfg.embed_dim) << (idx if side == "encoder" else nl - 1 - idx)
    # parameters of a block ~ 12 width^2; of the model ~ 2 sum_stages depth 12 width^2
    total = 2 * sum(int(d) * 12 * (int(cfg.embed_dim) << i) ** 2 for i, d in enumerate(cfg.depths))
    if depth >= 4 and depth * 12 * width * width * SPLIT_FRACTION > total:
        return depth // 2
    return 0


def stage_groups(cfg, stage_prefix: str) -> List[str]:
    """the group keys of one stage in announcement order: [prefix] or [prefix + "blocks[s:]", prefix + "blocks[:s]"]"""
    s = stage_split(cfg, stage_prefix)
    return [stage_prefix] if not s else [f"{stage_prefix}blocks[{s}:]", f"{stage_prefix}blocks[:{s}]"]


def backward_order_groups(cfg, skips_on_side: bool = True) -> List[str]:
    """Group keys in the order the backward announces them final: name prefixes, and for a stage announced in halves (stage_split)
    `<stage>blocks[s:]` (blocks s.. and the stage's other parameters) then `<stage>blocks[:s]`.  The ConvNeXt skip blocks' backward runs
    on the side stream beside the encoder stages (engine.skip_side), so their range is announced after the encoder's (before it with
    SCOT_SKIP_SIDE=0 / no side stream)."""
    nl = len(cfg.depths)
    g = ["patch_recovery."]
    for k in reversed(range(nl)):
        g += stage_groups(cfg, f"decoder.layers.{k}.")
    if not skips_on_side:
        g += ["residual_blocks."]
    for s in reversed(range(nl)):
        g += stage_groups(cfg, f"encoder.layers.{s}.")
    if skips_on_side:
        g += ["residual_blocks."]
    g += ["embeddings."]
    return g


def _in_group(name: str, key: str) -> bool:
    if "blocks[" not in key:
        return name.startswith(key)
    stage, sl = key.split("blocks[")
    if not name.startswith(stage):
        return False
    lo, hi = sl.rstrip("]").split(":")
    rest = name[len(stage):]
    if not rest.startswith("blocks."):
        return lo != ""                    # the stage's resampling layer goes with the upper half (its backward runs before the blocks')
    b = int(rest.split(".")[1])
    return (b >= int(lo)) if lo != "" else (b < int(hi))


def group_ranges(arena, groups: List[str]) -> List[Tuple[str, int, int]]:
    """Contiguous [start, end) element ranges of the arena covering each group (a name prefix or a stage half, see backward_order_groups)."""
    out = []
    for pre in groups:
        offs = [(arena.offsets[n], arena.offsets[n] + arena.numel(n)) for n in arena.shapes if _in_group(n, pre)]
        if not offs:
            continue
        out.append((pre, min(o[0] for o in offs), max(o[1] for o in offs)))
    return out


def native_init(dist=None, group=None, unique_id: Optional[bytes] = None, rank: Optional[int] = None, world: Optional[int] = None):
    """Bring up the communicator of the C ABI (`scot_dp_init`, csrc/dp.hip) once per process and return (world, rank).
    With a torch process group the 128-byte token travels through it (`broadcast_object_list` from rank 0) — the only use of
    torch.distributed on this path; a host without one passes `unique_id` (ops.dp_unique_id() of rank 0), `rank` and `world` itself."""
    from . import ops
    if ops.dp_world():
        return ops.dp_world(), ops.dp_rank()
    if unique_id is None:
        if dist is None:
            rank, world, unique_id = 0, 1, ops.dp_unique_id()
        else:
            rank, world = dist.get_rank(group), dist.get_world_size(group)
            box = [ops.dp_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            unique_id = box[0]
    ops.dp_init(unique_id, rank, world)
    return world, rank


class GradAllReducer:
    """backend: "torch" (the collectives of the process group `dist`: the default) or "native" (the C ABI's own RCCL communicator:
    scot_dp_init / scot_dp_allreduce_bucket, include/scot_hip.h — what a host without torch.distributed calls).
    wire: "fp32" (the reference's DDP semantics) or "bf16" (half the xGMI bytes: one HIP pass packs 1/N·g into bfloat16,
    the sum runs on the 16-bit buffer, one pass unpacks — scot_dp_pack / scot_dp_unpack).
    collective: "allreduce" (one RCCL all-reduce per chunk) or "rs_ag" (reduce-scatter + all-gather per chunk: the same ring
    traffic, but every rank owns the mean of 1/N of each chunk between the two halves — the seam a sharded optimizer step
    hangs on; backends without reduce_scatter_tensor (gloo, CPU tests) take the same bookkeeping through all_reduce)."""

    def __init__(self, model, dist, wire: str = "fp32", chunk_mb: int = 128, group=None, collective: str = "allreduce",
                 backend: str = "torch"):
        if wire not in ("fp32", "bf16") or collective not in ("allreduce", "rs_ag") or backend not in ("torch", "native"):
            raise ValueError("wire: fp32 | bf16; collective: allreduce | rs_ag; backend: torch | native")
        if backend == "native" and collective != "allreduce":
            raise ValueError("backend='native' exports the all-reduce only (scot_dp_allreduce_bucket)")
        self.model = model
        self.dist = dist
        self.group = group
        self.backend = backend
        self.world = dist.get_world_size(group) if dist is not None else 1
        self.rank = dist.get_rank(group) if dist is not None else 0
        if backend == "native":
            self.world, self.rank = native_init(dist, group)
        self.wire = wire
        self.collective = collective
        self.chunk = chunk_mb * (1 << 20) // 4
        self._wire_buf: Optional[torch.Tensor] = None
        self._pad_buf: Optional[torch.Tensor] = None
        self.comm_stream = None
        self.bytes_on_wire = 0          # per rank, payload handed to the collectives since construction

    def _flat(self) -> torch.Tensor:
        return self.model.flat_grads()

    def broadcast_parameters(self, src: int = 0):
        if self.backend == "native":       # the ABI exports ONE collective: a broadcast is the sum of the source's values and zeros
            from . import ops
            flat = self.model.flat_parameters()
            if self.rank != src:
                flat.zero_()
            ops.dp_allreduce(flat)
        else:
            self.dist.broadcast(self.model.flat_parameters(), src=src, group=self.group)
        # a collective writes the arena WITHOUT moving torch's version counter (checked on torch 2.10, gloo and nccl): without this the
        # engine of a non-source rank that has already run a forward would keep computing from 16-bit copies of its pre-broadcast weights
        self.model.mark_weights_dirty()

    def ranges_in_backward_order(self):
        return group_ranges(self.model._arena, backward_order_groups(self.model.config))

    def _sum(self, buf: torch.Tensor):
        """In-place sum over ranks of `buf` (a multiple of `world` elements long when collective == "rs_ag")."""
        self.bytes_on_wire += buf.numel() * buf.element_size()
        if self.backend == "native":
            from . import ops
            ops.dp_allreduce(buf)
        elif self.collective == "rs_ag":
            shard = buf.view(self.world, -1)[self.rank]
            if hasattr(self.dist, "reduce_scatter_tensor") and self.dist.get_backend(self.group) != "gloo":
                self.dist.reduce_scatter_tensor(shard, buf, group=self.group)        # this rank now owns its shard's sum
                self.dist.all_gather_into_tensor(buf, shard, group=self.group)
            else:   # same data flow on a backend without reduce-scatter: sum, keep the own shard, gather the shards
                self.dist.all_reduce(buf, group=self.group)
                parts = [torch.empty_like(shard) for _ in range(self.world)]
                self.dist.all_gather(parts, shard.clone(), group=self.group)
                for r, pt in enumerate(parts):
                    buf.view(self.world, -1)[r].copy_(pt)
        else:
            self.dist.all_reduce(buf, group=self.group)

    def reduce_range(self, start: int, end: int):
        """Mean all-reduce of arena[start:end] on the current stream."""
        from . import ops
        flat = self._flat()
        inv = 1.0 / self.world
        for s in range(start, end, self.chunk):
            e = min(end, s + self.chunk)
            seg = flat[s:e]
            n = e - s
            npad = (n + self.world * 8 - 1) // (self.world * 8) * (self.world * 8) if self.collective == "rs_ag" else n
            if self.wire == "bf16":
                if self._wire_buf is None or self._wire_buf.numel() < self.chunk + self.world * 8:
                    self._wire_buf = torch.zeros(self.chunk + self.world * 8, dtype=torch.bfloat16, device=flat.device)
                w = self._wire_buf[:npad]
                if npad != n:
                    w[n:].zero_()
                ops.dp_pack(seg, w, inv)                       # fp32 -> bf16 with the mean's 1/N, one HIP pass
                self._sum(w)                                   # sum on the wire
                ops.dp_unpack(w, seg, 1.0)                     # bf16 -> fp32, one HIP pass
            elif npad != n:
                if self._pad_buf is None or self._pad_buf.numel() < self.chunk + self.world * 8:
                    self._pad_buf = torch.zeros(self.chunk + self.world * 8, dtype=torch.float32, device=flat.device)
                w = self._pad_buf[:npad]
                w[n:].zero_()
                torch.mul(seg, inv, out=w[:n])
                self._sum(w)
                seg.copy_(w[:n])
            else:
                seg.mul_(inv)
                self._sum(seg)

    def allreduce(self):
        flat = self._flat()
        self.reduce_range(0, flat.numel())


class OverlappedGradAllReducer(GradAllReducer):
    """Launches each arena range's mean all-reduce on a side HIP stream as soon as the backward has finalised it
    (engine.on_grads_final), so that only the last range (the small C=96 encoder stage + embeddings) is exposed.
    Roughly 73 % of Poseidon-B's gradient bytes (the two C=768 stages) are final by mid-backward."""

    def __init__(self, model, dist, wire: str = "fp32", chunk_mb: int = 64, group=None, collective: str = "allreduce",
                 backend: str = "torch"):
        super().__init__(model, dist, wire=wire, chunk_mb=chunk_mb, group=group, collective=collective, backend=backend)
        self._ranges = None
        # a stream measured to overlap the engine's main chain and its weight-gradient stream (streams.py; created lazily in attach(): the
        # engine's side stream exists by then)
        self.comm_stream = None
        self._pending = False
        self._hooked = False
        self.timing = None      # set to [] to collect (prefix, start event, end event) per range on the comm stream

    def attach(self):
        if self.comm_stream is None:
            from .streams import forget, independent_stream
            eng = self.model._engine
            dev = self._flat().device
            if dev.type == "cuda":
                # the engine's weight-gradient stream and the communication stream are both picked AFTER the communicator exists (a reducer
                # is built on one): a weight-gradient stream older than the communicator can serialise against RCCL's internal streams
                # although every visible pair measures as overlapping — 21.6 vs 11.7 ms, streams.py "Communicators"
                forget(dev)
                eng.side = None
                self.comm_stream = independent_stream(dev, [torch.cuda.current_stream(dev), eng.side_stream() if eng.use_side else None])
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
        if self.comm_stream is None:      # no HIP streams (gloo on CPU tensors: the emulated tests): the range's exchange runs in line
            self.reduce_range(rng[0], rng[1])
            return
        ev = torch.cuda.Event()
        ev.record()
        self.comm_stream.wait_event(ev)
        with torch.cuda.stream(self.comm_stream):
            if self.timing is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self.reduce_range(rng[0], rng[1])
            if self.timing is not None:
                e1.record()
                self.timing.append((prefix, e0, e1))
        self._pending = True

    def comm_ms(self) -> float:
        """GPU time the collectives (incl. pack / unpack) took on the comm stream since `timing` was last reset."""
        t, self.timing = (self.timing or []), []
        return sum(a.elapsed_time(b) for _, a, b in t)

    def finish(self):
        if self._pending:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
            self._pending = False
