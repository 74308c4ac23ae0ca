"""ScOTEngine — the forward / backward program of the scOT hot path on MI355X.

Host-side orchestration only: every arithmetic operation below is one of the hand-written HIP kernels of
libscot_hip.so (poseidon_amd/ops.py → include/scot_hip.h); torch supplies device memory and the stream.
There is no CPU or eager-PyTorch fallback — a missing library raises.

The op sequence follows the reference graph (reference scOT/model.py:1318-1509) with the
data-flow choices described in DESIGN.md:
  * residual stream, LN statistics, softmax, q/k normalisation, gradient accumulation: fp32 in every mode;
  * compute="bf16": GEMM / attention operands are bf16 (activations between GEMMs are stored bf16),
    compute="fp32": everything fp32 with the exact fp32 MFMA (parity mode, ≤1e-5);
  * roll / window partition / reverse / mask are index math inside the attention kernel;
  * GELU is applied while loading the fc2 operand (the 4C activation is stored once, pre-activation);
  * gradients are accumulated in place into the flat gradient arena (autograd `+=` semantics).
"""
from __future__ import annotations

import math
import os
import weakref
from typing import Dict, List, Optional

import torch

from . import ops
from .arena import Arena
from .geometry import drop_path_rates, BlockGeom, StageGeom, stage_plan


def cpb_coords_table(ws: int) -> torch.Tensor:
    """relative_coords_table of HF:457-476 → [(2ws-1)^2, 2] (dy, dx), fp32 exactly as torch computes it."""
    r = torch.arange(-(ws - 1), ws, dtype=torch.int64).float()
    tab = torch.stack(torch.meshgrid([r, r], indexing="ij")).permute(1, 2, 0).contiguous()
    if ws > 1:
        tab = tab / (ws - 1)
    tab = tab * 8
    tab = torch.sign(tab) * torch.log2(torch.abs(tab) + 1.0) / math.log2(8)
    return tab.reshape(-1, 2).contiguous()


# Policy switches of the engine.  Every entry's losing setting has a committed measurement (profiles/HISTORY.md, profiles/round2..6), so none of
# them is an environment knob any more (round 6: 26 -> 5 `os.environ` reads in this file, one of them the generic SCOT_ENGINE_OPTIONS of the A/B scripts); tests and tools/ A/B scripts that still want the
# other setting patch this dict (or pass `options=` to ScOTEngine) before the engine is built.
ENGINE_OPTIONS = dict(
    tape_c=True,              # recorded steps are replayed inside the library (scot_tape_replay), not by a Python loop over the calls
    tape_inference=True,      # inference forwards are recorded / replayed too (rollouts: hundreds of forwards of one signature)
    skip_side=True,           # ConvNeXt skip blocks on the second stream, beside the deep stages' chain
    group_wgrads=True,        # a layer's weight gradients as one grouped launch
    cln_partial=True,         # small-row-count norm backward through per-workgroup partial sums
    lean_tail=True,           # fused layer tail without 4C-wide tensors in HBM (round 3: -0.25 ms)
    recycle=True,             # rows that die inside a layer come from a pool (stores land on lines the previous layer left in L2 / MALL)
    attn_x3=True,             # bf16x3 mode: split products inside the 16x16-window attention kernels too
    trunk_bf16=False,         # patch embed / merge / unmerge / recovery on 16-bit operands (costs ~1e-3 of output error each)
    trunk_x3=True,            # fp16 mode: the trunk on the split 16-bit MFMA (fp32 operands, 3 products)
    fused_mlp=True,           # csrc/mlp_fused.hip at C = 96 / 192 (round 2: 23.7 vs 24.9 ms)
    fused_tail=True,          # projection half + MLP half of a layer's tail in one launch per direction
    fused_min_rows=4096,      # ... from this many token rows on: a tail workgroup owns 64 / 128 rows, so 2048 rows are 32 workgroups on 256 CUs
                              # (round 6, profiles/round6/fused_min_rows_ab.txt: Poseidon-T batch 32 6.18 -> 5.72 ms, Poseidon-B batch 8 10.42 -> 9.41 ms;
                              #  16384 loses again: 5.85 — at 8192 rows the fused tail still wins)
    fused_fwd48=True,         # C = 48 (Poseidon-T / -S stage 0): the forward tail fused too (one 192-wide hidden chunk, padded-K MFMA steps)
    fused_bwd48=False,        # ... and the backward tail (the round-2 form: gelu'(u) stored, du stored, no qkv prologue; the lean form and
                              # scot_wgrad_mlp need C % 32 == 0): built and tested, measured NEUTRAL on Poseidon-T batch 32 (the weight-gradient
                              # stream paces that backward: 3.56 ms either way) — profiles/round6/tail48_ab.txt
    fused_next_qkv=(48, 96, 192),   # widths at which the forward tail also produces the next layer's q/k/v projection
    fused_qkv_dgrad=(96,),      # widths at which the backward tail applies the previous layer's qkv data gradient as a prologue (192 spills)
    dgrad_wt=True,            # transposed 16-bit weight copies: data gradients as NT products (stages 2/3: 1.7-2.2x)
    grad_scale="auto",        # fp16 mode: initial power-of-two gradient scale ("auto" = from the loss normalisation; a number; "1" = off)
    ls_rescale=True,          # fp16 mode: ConvNeXt skip branches run their backward under an extra power of two (layer scale ~1e-6)
    lazy_grads=True,          # first writers of the ScOTLayers' weight gradients store (round 6: 18.58 -> 18.24 ms)
    attn_rep=16,              # replicas of the attention backward's atomically accumulated table / logit-scale gradients
)


class _Token:
    """Lifetime marker of one taped forward call: alive while the caller (the autograd node) can still ask for its backward."""
    __slots__ = ("__weakref__",)


class ScOTEngine:
    def __init__(self, cfg, arena: Arena, compute: str = "fp16", options: Optional[dict] = None):
        if compute not in ("fp16", "bf16", "fp32", "bf16x3"):
            raise ValueError("compute must be 'fp16', 'bf16', 'fp32' or 'bf16x3'")
        opt = dict(ENGINE_OPTIONS, **(options or {}))
        for kv in filter(None, os.environ.get("SCOT_ENGINE_OPTIONS", "").split(",")):      # A/B runs only (tools/gpu_ab.sh): "lazy_grads=0,attn_rep=8"
            k, _, v = kv.partition("=")
            cur = ENGINE_OPTIONS.get(k.strip())
            opt[k.strip()] = (v.strip() not in ("0", "false", "False")) if isinstance(cur, bool) else \
                (int(v) if isinstance(cur, int) else (tuple(int(c) for c in v.split("+") if c) if isinstance(cur, tuple) else v.strip()))
        if set(opt) != set(ENGINE_OPTIONS):
            raise ValueError(f"unknown engine option(s): {sorted(set(opt) - set(ENGINE_OPTIONS))}")
        self.options = opt
        # "fp16" and "bf16" run the SAME kernels from two builds of the library (csrc/common.h: the format of the 16-bit operand
        # type is a compile-time property); everything else about the two modes is identical except the backward's gradient scale
        self.lib_kind = "f16" if compute == "fp16" else "bf16"
        half = compute in ("fp16", "bf16")
        self.cfg = cfg
        self.stage_timing = os.environ.get("SCOT_STAGE_TIMING", "0") == "1"
        self.marks = []
        # step tape: the second training step with a given input signature is recorded (every C-ABI launch with its final
        # arguments + the host-side stream/event operations between them), later steps replay the list: ~2500 launches per
        # step cost ~10 us of Python each when issued through the op wrappers, ~1.5 us when replayed.
        self.tape_mode = os.environ.get("SCOT_TAPE", "1") == "1"
        self.stochastic = False
        self.launch_timer = None    # bench.py: list that collects per-launch HIP-event timings of replayed steps
        self.tape_max = max(1, int(os.environ.get("SCOT_TAPE_MAX", "2")))
        # ... and replayed from C: one scot_tape_replay call per run of launches instead of one ctypes call per launch (option tape_c=False: the
        # Python loop over the recorded calls)
        self.tape_c = opt["tape_c"] and __import__("platform").machine() in ("x86_64", "AMD64")     # (scot_tape_replay is System V x86-64 only)
        self.tape_inference = opt["tape_inference"]      # inference forwards are recorded / replayed too
        self._rec = None
        self._rec_keep = None
        self._taped = {}
        self._pending = []
        self._wgq = []                                                  # weight gradients waiting to be grouped (see wgrad)
        self._finq = []                                                 # per-workgroup partial sums of norm backwards waiting for their column sums (see finish_partials)
        self._in_side = None                                            # main stream while a side-stream task runs (see fork_task)
        self._task_keep, self._task_keeps = [], {}
        self._events = []                                               # events of the current step (a recorded step keeps its own alive)
        # ConvNeXt skip blocks off the critical path: a skip's blocks only feed the decoder stage that consumes the skip (forward)
        # / the encoder stage that produced it (backward), so they run on the side stream beside the deep stages' latency-bound
        # chain instead of in front of it (option skip_side=False: in line)
        self.skip_side = opt["skip_side"]
        self.group_wgrads = opt["group_wgrads"]
        self.grad_fill_event = None       # ScOT.zero_grad(overlap=True): recorded behind the gradient arena's fill on the side stream
        self.cln_partial = opt["cln_partial"]     # small-row-count LN backward through partial sums
        # the fused layer tail WITHOUT 4C-wide tensors in HBM (round 3): the forward stores neither gelu(u) nor gelu'(u) and keeps the
        # pre-norm rows as 16-bit, the backward recomputes gelu'(u) and does not store du, scot_wgrad_mlp recomputes both for the
        # fc1 / fc2 weight gradients, the norms' parameter gradients go through per-workgroup partial rows instead of atomics
        self.lean_tail = opt["lean_tail"]
        # Rows that are dead before a layer ends (the fp32 residual h between a layer's two halves) or once the NEXT layer has read them
        # (its fp32 output; in inference every intermediate) come from a small pool keyed by (tag, shape) instead of fresh memory: a
        # recorded step owns every buffer it allocated for good, so without the pool each layer's stores go to lines no cache has
        # seen — with it they land on lines the previous layer left in L2 / MALL.
        self.recycle = opt["recycle"]
        self.last_hidden, self.last_hidden_aliased = None, False
        self._pool: Dict[tuple, torch.Tensor] = {}
        # ... gelu'(u) itself IS stored (16-bit, 8·C bytes per token) and the backward tail loads it: recomputing it there (the C ABI's
        # `dact = NULL` form of scot_block_tail_bwd) is one more C x 4C product per row tile at the 256-register cap — 140-168 B/lane of
        # scratch, +20 us per launch, +0.3 ms per step (round 3) — so the engine no longer offers it
        self.arena = arena
        # bf16x3: activations and weights stay fp32 in HBM; the GEMMs split them into hi + lo bf16 while staging into LDS and
        # run three bf16 MFMAs per K-step (≈ fp32 accuracy at the bf16 MFMA rate); so do the 16x16-window attention kernels
        self.compute = {"fp16": ops.BF16, "bf16": ops.BF16, "fp32": ops.F32, "bf16x3": ops.X3}[compute]
        # attention kernels' arithmetic: bf16x3 also splits inside the 16x16-window kernels (option attn_x3=False: exact fp32 MFMA)
        self.acm = ops.BF16 if half else (ops.X3 if (compute == "bf16x3" and opt["attn_x3"]) else ops.F32)
        self.adt = ops.HALF[self.lib_kind] if half else torch.float32
        self.device = arena.data.device
        # The "trunk" (patch embed, merge, unmerge, recovery: < 2 % of the FLOPs) is the only path every output pixel's
        # signal must traverse; each bf16 GEMM on it adds ~1e-3 of relative error that nothing downstream averages out.
        # It therefore always runs on the exact fp32 MFMA with fp32 operands (option trunk_bf16 restores 16-bit operands).
        trunk32 = self.compute == ops.BF16 and not opt["trunk_bf16"]
        self.tcm = ops.F32 if (trunk32 or self.compute != ops.BF16) else ops.BF16
        # ... on the SPLIT 16-bit MFMA in the fp16 mode (fp32 operands, hi + lo bfloat16 halves, three MFMAs: 2^-17 operand error, 1/5 of
        # the exact fp32 MFMA's time; ops.gemm routes it to the bfloat16 build, whose halves keep fp32's range under the gradient scale).
        # option trunk_x3=False: exact fp32 MFMA
        if compute == "fp16" and self.tcm == ops.F32 and opt["trunk_x3"]:
            self.tcm = ops.X3
        # (tried in round 2: the trunk on the split 16-bit MFMA instead of the exact fp32 MFMA — 0.19 ms faster, same forward
        # parity, but the fp32 trunk gradients under the fp16 build's gradient scale exceed binary16's range in the split: NaN)
        self.tadt = torch.float32 if self.tcm != ops.BF16 else self.adt
        self.grid, self.enc, self.dec = stage_plan(cfg)
        self.drop_rates = drop_path_rates(cfg)      # per-layer stochastic-depth rate (0 for the training recipe, train.py:262)
        self.precision_probe = None                 # tools/probes: set of layer pieces run in fp32 during an inference forward
        self.drop_path_masks = None                 # tests: {(layer prefix, branch 0|1): [B] scale} instead of random draws
        self.cond = bool(cfg.use_conditioning)
        self._coords: Dict[int, torch.Tensor] = {}
        self._loss_meta = None
        # called with a state-dict prefix ("patch_recovery.", "decoder.layers.3.", …) as soon as the backward has FINISHED
        # writing every gradient under that prefix — the data-parallel wrapper launches that range's all-reduce there
        self.on_grads_final = None
        # Weight gradients are off the backward's critical path (nothing downstream reads dW): they are launched on a
        # second HIP stream, forked from / joined into the main stream with events, so that the (latency-bound) wgrad GEMMs
        # fill the CUs the dgrad / LN / attention chain leaves idle.  SCOT_SIDE_STREAM=0 serialises everything.
        self.use_side = os.environ.get("SCOT_SIDE_STREAM", "1") != "0" and torch.device(self.device).type == "cuda"
        self.side = None
        self._keep = []
        # csrc/mlp_fused.hip (validated and measured on MI355X in round 2: 23.7 vs 24.9 ms/step): fc1 → GELU → fc2 → cond-LN →
        # residual in one launch (and its backward chain, and the projection + LN pair) for the C = 96 / 192 stages of the 16-bit
        # modes; option fused_mlp=False restores the layer-by-layer launches
        self.fused_mlp = opt["fused_mlp"] and half
        self.fused_tail = opt["fused_tail"]     # MLP-half + projection-half backward in one launch
        self.fused_min_rows = int(opt["fused_min_rows"])
        # ... forward: the next layer's q/k/v projection as epilogue (channel widths).  Backward: the previous layer's qkv dgrad as
        # prologue — at C = 96 only: the C = 192 prologue variant spills and costs more than the GEMM it replaces (125 vs 87 + 20 us)
        self.fused_next_qkv = set(opt["fused_next_qkv"])
        self.fused_qkv_dgrad = set(opt["fused_qkv_dgrad"])
        # bf16 mode: GEMM operands must already be bf16 in HBM (gemm_fast streams raw 16-byte chunks into LDS), so the
        # weights get a bf16 shadow arena that is re-cast from the fp32 master at the start of EVERY forward (one pass,
        # inside the timed step), and every producer of a GEMM operand also writes a bf16 copy.
        self.shadow = torch.empty(arena.size, dtype=self.adt, device=self.device) if self.compute == ops.BF16 else None
        # ... and a second copy holding every weight MATRIX transposed (same offsets): the data gradients dX = dY · W then run as the
        # forward's NT product on W^T instead of the strided-operand NN product (stages 2/3: 1.7–2.2x slower for the same shape).
        # Filled by one launch per training forward, on the side stream (scot_transpose_cast).  option dgrad_wt=False: NN products.
        self.shadow_t, self._wt_names, self._wt_desc, self._wt_tiles = None, {}, None, 0
        if self.shadow is not None and opt["dgrad_wt"]:
            self._plan_transposed_weights()
        # fp16 operands have 5 exponent bits: the backward runs on gradients multiplied by a power of two chosen from the loss
        # normalisation (d loss / d prediction = O(1 / number of output elements); see _grad_scale) and the gradient arena is
        # divided by it afterwards (exact; scot_scale_inplace also counts non-finite values → `grad_overflow`).
        self.scale_grads = compute == "fp16" and str(opt["grad_scale"]) != "1"
        self.grad_overflow = torch.zeros(1, dtype=torch.int32, device=self.device) if self.scale_grads else None
        # {S, 1/S, applied steps since S last changed}: ON THE DEVICE, so that a recorded step never bakes a value in and the optimizer
        # (scot_optim_finish: torch.cuda.amp.GradScaler's rule — halve after an overflowed step, double after N clean ones) can change
        # it between steps.  Initialised from the loss normalisation by the first training forward (_init_grad_scale).
        self.scale_state = torch.tensor([1.0, 1.0, 0.0, 0.0], dtype=torch.float32, device=self.device) if self.scale_grads else None
        self._scale_ready = False
        self.grads_are_zero = False   # set by ScOT.zero_grad / _prepare_grads: the arena needs no pre-scaling then
        # Lazy zero-grad (16-bit modes): the Linear weights of the ScOTLayers — 95 % of the gradient bytes — are written by two kernel
        # families only (scot_wgrad_group, scot_wgrad_mlp), which can STORE `acc / S` instead of adding into a zero-filled tensor: after
        # ScOT.zero_grad(lazy=True) only the rest of the arena (`_small_chunks`) is cleared, the first backward's weight gradients are
        # stores that also carry the fp16 un-scale, and the un-scale pass only visits the rest.  Per step of Poseidon-B that is 600 MB not
        # filled, 600 MB of zeros not re-read, 1.2 GB not read + rewritten by the un-scale.  Later backwards of an accumulation window add
        # `acc / S` (the tensors hold unscaled values).  autograd's accumulate-into-.grad semantics: reference trainer.py:605-635.
        self.lazy_grads = False       # set by ScOT.zero_grad(lazy=True), cleared by the backward that consumed it
        self._lazy_now = False        # ... as seen by the backward in flight (a recorded backward exists per value)
        self._big_ptrs, self._small_chunks, self._small_by_key = set(), None, {}
        self.collect_attn, self.attn_sink = False, []   # output_attentions: one probability tensor per stage (encoder stages first)
        # ... and one global scale cannot also lift the gradients of a branch behind a ~1e-6 layer scale (2^-20 below the rest):
        # the ConvNeXt skip blocks run their backward under an extra, device-side power of two (convnext_bwd)
        self._gredirect, self._ls = None, {}
        if self.scale_grads and opt["ls_rescale"]:
            self._plan_layer_scale_rescale()
        self._wviews: Dict[str, torch.Tensor] = {}
        self._lean_cache: Dict[tuple, bool] = {}
        # 16-bit weight copies are refreshed only when the fp32 master changed: `weights_version()` (set by ScOT: in-place edits of
        # the parameters / the arena and the fused optimizer's steps all move it) is compared with the version the copies were made
        # from.  None (an engine built by hand): refresh at every forward.  The fused AdamW writes the copies itself.
        self.weights_version = None
        self._shadow_v = self._shadow_t_v = object()
        self._copies_maintained = False      # True once an optimizer (FusedAdamW) writes the 16-bit copies itself
        self._build_cpb_plan()
        if self.compute == ops.BF16 and arena.grad is not None and opt["lazy_grads"]:
            self._plan_grad_partition()

    def _build_cpb_plan(self):
        """All layers' continuous-position-bias MLPs run as ONE batched launch per step (forward) and one per stage
        (backward, so that a stage's gradients are final before its all-reduce)."""
        blocks = [b for st in self.enc for b in st.blocks] + [b for st in self.dec for b in st.blocks]
        wss = sorted({b.window_shift()[0] for b in blocks})
        coords, coff, cur = [], {}, 0
        for ws in wss:
            c = cpb_coords_table(ws)
            coff[ws] = cur
            cur += c.numel()
            coords.append(c.reshape(-1))
        self.cpb_coords = torch.cat(coords).to(self.device)
        desc, tab_off, z_off = [], 0, 0
        self.cpb_index: Dict[str, int] = {}
        self.cpb_slices: Dict[str, tuple] = {}
        for i, b in enumerate(blocks):
            ws = b.window_shift()[0]
            ts = (2 * ws - 1) ** 2
            a = b.prefix + ".attention.self.continuous_position_bias_mlp."
            o = self.arena.offsets
            desc += [o[a + "0.weight"], o[a + "0.bias"], o[a + "2.weight"], coff[ws], ws, b.heads, tab_off, z_off]
            self.cpb_index[b.prefix] = i
            self.cpb_slices[b.prefix] = (tab_off, b.heads, ts)
            tab_off += b.heads * ts
            z_off += b.heads * ts
        self.cpb_desc = torch.tensor(desc, dtype=torch.int32).to(self.device)
        self.cpb_nlayers = len(blocks)
        self.cpb_max_ws = max(wss)
        self.cpb_tables = torch.empty(tab_off, device=self.device)
        self.cpb_z = torch.empty(z_off, device=self.device)
        # the attention backward accumulates dtable / dlogit_scale with atomics: R replicas (window w -> replica w % R) keep the
        # same-address chains short; replica 0 of the tables is what the bias-MLP backward reads after the per-stage fold
        self.attn_rep = R = max(1, int(self.options["attn_rep"]))
        self.cpb_tab_total, self.cpb_ls_total = tab_off, sum(b.heads for b in blocks)
        self.cpb_dtables = torch.zeros(R * tab_off, device=self.device)
        self.cpb_dls = torch.zeros(R * self.cpb_ls_total, device=self.device) if R > 1 else None
        self.cpb_ls_off, cur = {}, 0
        for b in blocks:
            self.cpb_ls_off[b.prefix] = cur
            cur += b.heads
        self._rep_desc = {}

    def cpb_table(self, prefix, grad=False):
        off, heads, ts = self.cpb_slices[prefix]
        return (self.cpb_dtables if grad else self.cpb_tables)[off:off + heads * ts].view(heads, ts)

    # ------------------------------------------------------------------------------------------ gradient arena: stored part / filled part
    def _plan_grad_partition(self):
        """`_big_ptrs`: addresses of the gradient tensors whose first writer stores (q/k/v as one [3C, C] span, attention.output.dense,
        intermediate.dense, output.dense weights of every ScOTLayer); `_small_chunks`: the rest of the arena as (offset, count <= 4096)
        pieces for scot_segments_scale."""
        ar = self.arena
        big = []
        for blk in [b for st in self.enc for b in st.blocks] + [b for st in self.dec for b in st.blocks]:
            pre, C = blk.prefix, blk.dim
            hid = int(self.cfg.mlp_ratio * C)
            for name, n in ((pre + ".attention.self.qkv_weight", 3 * C * C), (pre + ".attention.output.dense.weight", C * C),
                            (pre + ".intermediate.dense.weight", hid * C), (pre + ".output.dense.weight", C * hid)):
                o = ar.offsets[name]
                if o % 64 or n % 64:
                    continue
                big.append((o, o + n))
                self._big_ptrs.add(ar.grad.data_ptr() + 4 * o)
        big.sort()
        small, cur = [], 0
        for lo, hi in big:
            if lo > cur:
                small.append((cur, lo))
            cur = max(cur, hi)
        if cur < ar.size:
            small.append((cur, ar.size))
        self._small_segs = small
        self._small_chunks = self._chunk_tensor(small)

    def _chunk_tensor(self, segs):
        rows = []
        for lo, hi in segs:
            for o in range(lo, hi, 4096):
                rows += [o, min(4096, hi - o)]
        t = torch.tensor(rows, dtype=torch.int64).reshape(-1, 2).to(self.device) if rows else torch.zeros(0, 2, dtype=torch.int64, device=self.device)
        return t, t.shape[0]

    def small_chunks(self, key=None):
        """(descriptor tensor, count) of the zero-filled / un-scaled part of the gradient arena, whole or inside one announced range"""
        if key is None:
            return self._small_chunks
        c = self._small_by_key.get(key)
        if c is None:
            from .dp import group_ranges
            segs = []
            for _, lo, hi in group_ranges(self.arena, [key]):
                hi = min(self.arena.size, (hi + 63) // 64 * 64)      # (a range ends with its last tensor's last element; the alignment gap behind it is nobody's)
                segs += [(max(a, lo), min(b, hi)) for a, b in self._small_segs if max(a, lo) < min(b, hi)]
            c = self._small_by_key[key] = self._chunk_tensor(segs)
        return c

    def fill_small_grads(self):
        """zero the part of the gradient arena that is accumulated into (ScOT.zero_grad(lazy=True))"""
        prev = ops.use(self.lib_kind)
        try:
            ops.segments_scale(self.arena.grad, self._small_chunks[0], self._small_chunks[1], None)
        finally:
            ops.use(prev)

    def grad_mode(self, gw) -> int:
        """how a weight gradient meets its tensor in the backward in flight (ops.GRAD_*)"""
        if gw.data_ptr() not in self._big_ptrs:
            return ops.GRAD_ADD
        if self._lazy_now:
            return ops.GRAD_STORE_SCALED
        return ops.GRAD_ADD_SCALED if self.scale_grads else ops.GRAD_ADD

    def grad_unscale(self):
        return self.scale_state[1:2] if self.scale_grads else None

    def scale_grad_range(self, factor_dev, key=None, count=False):
        """the arena (or one announced range of it) to / from the gradient scale: only the part the scale is not folded into"""
        ov = self.grad_overflow if count else None
        if self._small_chunks is not None:
            ch, n = self.small_chunks(key)
            ops.segments_scale(self.arena.grad, ch, n, factor_dev, ov)
            return
        if key is None:
            ops.scale_inplace_dev(self.arena.grad, factor_dev, ov)
            return
        from .dp import group_ranges
        for _, lo, hi in group_ranges(self.arena, [key]):
            ops.scale_inplace_dev(self.arena.grad[lo:hi], factor_dev, ov)

    def cpb_backward_range(self, blocks):
        """Bias-MLP backward of a stage's layers (reads the table gradients the attention backward accumulated, writes only
        parameter gradients): ~100 us per call that nothing downstream waits for -> side stream."""
        if blocks:
            first, n = self.cpb_index[blocks[0].prefix], len(blocks)
            mw, mh = max(b.window_shift()[0] for b in blocks), max(b.heads for b in blocks)
            def run():
                if self.attn_rep > 1:
                    td, sd, tmax, smax = self._replica_desc(blocks)
                    ops.replica_reduce(self.cpb_dtables, 1, self.attn_rep, self.cpb_tab_total, td, 1, tmax, self.cpb_dtables)
                    ops.replica_reduce(self.cpb_dls, 0, self.attn_rep, self.cpb_ls_total, sd, n, smax, self.arena.grad)
                ops.cpb_bwd_batched(self.arena.data, self.cpb_desc, first, n, mw, mh, self.cpb_coords, self.cpb_z, self.cpb_dtables,
                                    self.arena.grad)
            self.off_critical_path(run)
            self.flush_side()

    def _replica_desc(self, blocks):
        """(src_off, dst_off, count) entries that fold a stage's table-gradient replicas into replica 0 and its logit-scale replicas
        into the arena's gradients."""
        key = blocks[0].prefix
        d = self._rep_desc.get(key)
        if d is None:
            lo = self.cpb_slices[blocks[0].prefix][0]
            off, heads, ts = self.cpb_slices[blocks[-1].prefix]
            hi = off + heads * ts
            sd = []
            for b in blocks:
                sd += [self.cpb_ls_off[b.prefix], self.arena.offsets[b.prefix + ".attention.self.logit_scale"], b.heads]
            d = self._rep_desc[key] = (torch.tensor([lo, lo, hi - lo], dtype=torch.int32).to(self.device),
                                       torch.tensor(sd, dtype=torch.int32).to(self.device), hi - lo, max(b.heads for b in blocks))
        return d

    # ------------------------------------------------------------------------------------------ helpers
    def P(self, name):
        return self.arena.view(name)

    def G(self, name):
        r = self._gredirect.get(name) if self._gredirect is not None else None
        return r if r is not None else self.arena.gview(name)

    def _plan_layer_scale_rescale(self):
        """fp16 mode: every ConvNeXt skip block gets a scratch copy of its range of the gradient arena and a device-side pair
        (c, 1/c); see convnext_bwd and csrc/misc.hip (scot_pow2_rescale)."""
        ar = self.arena
        blocks = {}
        for name in ar.shapes:
            if name.startswith("residual_blocks.") and name.count(".") >= 3:
                blocks.setdefault(".".join(name.split(".")[:3]), []).append(name)
        for pre, names in blocks.items():
            if pre + ".weight" not in names or pre + ".pwconv2.weight" not in names:
                continue        # not a ConvNeXt block
            lo = min(ar.offsets[n] for n in names)
            hi = max(ar.offsets[n] + ar.numel(n) for n in names)
            if any(lo <= o < hi for n, o in ar.offsets.items() if n in ar.shapes and not n.startswith(pre + ".")):
                continue        # (cannot happen with the registration-order layout: a block's parameters are contiguous)
            scratch = torch.zeros(hi - lo, dtype=torch.float32, device=self.device)
            views = {n: scratch[ar.offsets[n] - lo: ar.offsets[n] - lo + ar.numel(n)].view(ar.shapes[n]) for n in names if n != pre + ".weight"}
            self._ls[pre] = dict(lo=lo, hi=hi, scratch=scratch, views=views, cs=torch.ones(2, dtype=torch.float32, device=self.device))

    def W(self, name):
        """Weight `name` as a GEMM operand (compute dtype)."""
        if self.shadow is None:
            return self.arena.view(name)
        v = self._wviews.get(name)
        if v is None:
            o = self.arena.offsets[name]
            v = self.shadow[o:o + self.arena.numel(name)].view(self.arena.shapes[name])
            self._wviews[name] = v
        return v

    def Wspan(self, name, numel):
        o = self.arena.offsets[name]
        return (self.shadow if self.shadow is not None else self.arena.data)[o:o + numel]

    def _plan_transposed_weights(self):
        ar = self.arena
        mats = []
        for name, off in ar.offsets.items():
            if name.endswith(".qkv_weight"):
                C = ar.shapes[name[: -len("qkv_weight")] + "query.weight"][0]
                mats.append((name, off, 3 * C, C))
            elif name.endswith("weight") and name in ar.shapes and len(ar.shapes[name]) == 2 and \
                    not name.endswith(("query.weight", "key.weight", "value.weight")):
                r, c = ar.shapes[name]
                if r % 8 == 0 and c % 8 == 0 and min(r, c) >= 32:
                    mats.append((name, off, r, c))
        if not mats:
            return
        desc, tile = [], 0
        for name, off, r, c in mats:
            desc.append((off, r, c, tile))
            tile += ((r + 63) // 64) * ((c + 63) // 64)
            self._wt_names[name] = (off, r, c)
        self.shadow_t = torch.zeros(ar.size, dtype=self.adt, device=self.device)
        self._wt_desc = torch.tensor(desc, dtype=torch.int32, device=self.device)
        self._wt_tiles = tile
        self._wtviews = {}

    def transpose_weights(self):
        ops.transpose_cast(self.arena.data, self.shadow_t, self._wt_desc, len(self._wt_names), self._wt_tiles)

    def WT(self, name, w=None):
        """W(name)^T as a GEMM operand ([in, out], contiguous), or None when no transposed copy is kept (`w`: the operand the caller
        is about to use — a transposed copy only stands in for the 16-bit weight copy, never for the fp32 master)."""
        if self.shadow_t is None or name not in self._wt_names or (w is not None and w.dtype != self.adt):
            return None
        v = self._wtviews.get(name)
        if v is None:
            off, r, c = self._wt_names[name]
            v = self.shadow_t[off:off + r * c].view(c, r)
            self._wtviews[name] = v
        return v

    def TW(self, name):
        """Trunk weight operand (fp32 master when the trunk computes in fp32)."""
        return self.arena.view(name) if self.tcm != ops.BF16 else self.W(name)

    def to_tadt(self, x):
        if self.tadt == torch.float32:
            return x
        return self.to_adt(x)

    def to_adt(self, x):
        """Copy of fp32 `x` in the GEMM operand dtype (identity in fp32 mode)."""
        if self.adt == torch.float32:
            return x
        y = self.new(*x.shape, dtype=self.adt)
        ops.cast(x, y)
        return y

    def new(self, *shape, dtype=torch.float32):
        if self._in_side is not None:
            # a side-stream task that produces tensors the main chain consumes later (skip blocks): allocate from the MAIN
            # stream's pool, so the block's reuse stays ordered with the main-stream kernels that read it after the join
            with torch.cuda.stream(self._in_side):
                t = torch.empty(*shape, dtype=dtype, device=self.device)
            self._task_keep.append(t)      # ... and alive until the main stream has waited for the task (wait_task)
        else:
            t = torch.empty(*shape, dtype=dtype, device=self.device)
        if self._rec is not None:
            self._rec_keep.append(t)   # a recorded step owns its buffers for good: replays reuse these very addresses
        return t

    def pool(self, tag, *shape, dtype=torch.float32):
        """a buffer nobody reads after the consumer the caller is about to launch (see `recycle`): one allocation per (tag, shape, dtype)"""
        if not self.recycle or self._in_side is not None:
            return self.new(*shape, dtype=dtype)
        key = (tag, shape, dtype, self._stream_id())      # (forwards issued from two streams must not share rows)
        t = self._pool.get(key)
        if t is None:
            t = self._pool[key] = torch.empty(*shape, dtype=dtype, device=self.device)
        if self._rec is not None:
            self._rec_keep.append(t)      # (a recorded step names the address: it keeps the buffer alive whatever happens to the pool)
        return t

    def zeros(self, *shape, dtype=torch.float32):
        t = self.new(*shape, dtype=dtype)
        self.h_zero(t)
        return t

    # Host-side operations between launches, as C-ABI calls on the GPU (so that they are ordinary tape entries and a recorded forward or
    # backward replays as one run inside the library); torch calls on the CPU emulation of the tests.
    def _native_host_ops(self):
        return self.device.type == "cuda"

    def h_zero(self, t):
        if self._native_host_ops() and t.is_contiguous():
            ops.memset_async(t, 0)
        else:
            self.tdo(t.zero_)

    def h_copy(self, dst, src):
        if self._native_host_ops() and dst.dtype == src.dtype and dst.is_contiguous() and src.is_contiguous() and dst.numel() == src.numel():
            ops.memcpy_async(dst, src)
        else:
            self.tdo(lambda: dst.copy_(src))

    def _event(self):
        """a torch event whose HIP handle exists (torch creates it at the first record), alive as long as the launches that name it: for
        good inside a recorded step, until the next forward otherwise"""
        ev = torch.cuda.Event()
        ev.record()
        (self._rec_keep if self._rec is not None else self._events).append(ev)
        return ev

    def h_record(self, ev, stream):
        if self._native_host_ops():
            ops.event_record(ev.cuda_event, stream.cuda_stream)
        else:
            self.tdo(lambda: ev.record(stream))

    def h_wait(self, stream, ev):
        if self._native_host_ops():
            ops.stream_wait_event(stream.cuda_stream, ev.cuda_event)
        else:
            self.tdo(lambda: stream.wait_event(ev))

    def tdo(self, fn):
        """Host-side operation that is part of the step but not a C-ABI launch (torch memset/copy, event record, stream
        wait, DP callback): run it, and log it when a step tape is being recorded."""
        fn()
        if self._rec is not None:
            self._rec.append((fn, None))

    def tdo_dynamic(self, fn):
        """Like tdo, for a host-side step that decides AT RUN TIME which launches to issue (so the launches themselves must not
        be logged by the recorder: a replay calls fn again)."""
        def run():
            prev = ops.set_recorder(None)
            try:
                fn()
            finally:
                ops.set_recorder(prev)
        run()
        if self._rec is not None:
            self._rec.append((run, None))

    def _grad_scale(self, n_out: int) -> float:
        """INITIAL power-of-two factor the fp16 backward runs under.  d loss / d prediction is O(1 / n_out) for the (relative) mean
        losses of model.py:1424-1484, i.e. 2.4e-7 for Poseidon-B at batch 64 — a subnormal in binary16; with the scale the
        gradient of the prediction is O(1 / mean|label|) and the 16-bit gradient tensors of the backward (dY operands) sit in
        the middle of binary16's 30 binades."""
        if not self.scale_grads:
            return 1.0
        env = str(self.options["grad_scale"])
        if env != "auto":
            return float(env)
        return float(2 ** max(0, int(math.floor(math.log2(max(1, n_out))))))

    def _init_grad_scale(self, n_out: int):
        """first training forward: scale_state <- the automatic choice (later changes are the optimizer's, on the device)"""
        if self.scale_grads and not self._scale_ready:
            S = self._grad_scale(n_out)
            self.scale_state.copy_(torch.tensor([S, 1.0 / S, 0.0, 0.0]))
            self._scale_ready = True

    def grad_scale_value(self) -> float:
        """current gradient scale (a host read: synchronises)"""
        return float(self.scale_state[0]) if self.scale_grads else 1.0

    def clone(self, t):
        y = self.new(*t.shape, dtype=t.dtype)
        self.h_copy(y, t)
        return y

    def coords(self, ws: int) -> torch.Tensor:
        t = self._coords.get(ws)
        if t is None:
            t = cpb_coords_table(ws).to(self.device)
            self._coords[ws] = t
        return t

    def _norm_params(self, prefix):
        if self.cond:
            return (self.P(prefix + ".weight.weight"), self.P(prefix + ".weight.bias"), self.P(prefix + ".bias.weight"),
                    self.P(prefix + ".bias.bias"))
        return (None, self.P(prefix + ".weight"), None, self.P(prefix + ".bias"))

    def _norm_grads(self, prefix):
        if self.cond:
            return (self.G(prefix + ".weight.weight"), self.G(prefix + ".weight.bias"), self.G(prefix + ".bias.weight"),
                    self.G(prefix + ".bias.bias"))
        return (None, self.G(prefix + ".weight"), None, self.G(prefix + ".bias"))

    def norm_fwd(self, prefix, x, resid, rows_per_sample, C, eps, time, out_dtype=torch.float32, need_stats=True, copy=False,
                 sample_scale=None, out=None, out16=None):
        """→ (out, out16, stats); out16 = operand-dtype copy of out (only when copy=True; == out in fp32 mode).  out / out16: buffers to
        write instead of fresh ones (the pooled rows of layer_fwd)."""
        rows = x.numel() // C
        if out is None:
            out = self.new(rows, C, dtype=out_dtype)
        if copy:
            if self.adt == torch.float32 or out_dtype == self.adt:
                out16 = out
            elif out16 is None:
                out16 = self.new(rows, C, dtype=self.adt)
        else:
            out16 = None
        mean = self.new(rows) if need_stats else None
        rstd = self.new(rows) if need_stats else None
        gw_w, gw_b, bw_w, bw_b = self._norm_params(prefix)
        ops.cln_fwd(x, resid, out, mean, rstd, time if self.cond else None, gw_w, gw_b, bw_w, bw_b, rows, rows_per_sample, C, eps,
                    out2=out16 if (out16 is not None and out16 is not out) else None, sample_scale=sample_scale)
        return out, out16, (mean, rstd)

    def norm_bwd(self, prefix, dout, x, stats, rows_per_sample, C, time, dx_dtype, sample_scale=None):
        rows = x.numel() // C
        dx = self.new(rows, C, dtype=dx_dtype)
        gw_w, gw_b, _, _ = self._norm_params(prefix)
        g = self._norm_grads(prefix)
        t = time if self.cond else None
        nf = ops.cln_bwd_partial_floats(rows, rows_per_sample, C, self.cond) if self.cln_partial else 0
        if nf and self._norm_grads_contiguous(prefix, C):
            # the deep stages' norms (1024 / 4096 rows): dx on the chain with every row's loads in flight at once, the cross-block
            # sums of the parameter gradients through a small partial matrix finished on the weight-gradient stream
            part = self.new(nf)
            ops.cln_bwd(dout, x, stats[0], stats[1], t, gw_w, gw_b, dx, None, None, None, None, rows, rows_per_sample, C,
                        sample_scale=sample_scale, mode=3, partial=part)
            ncol = (4 if self.cond else 2) * C
            self._finq.append((part, nf // ncol, ncol, next(t_ for t_ in g if t_ is not None)))
            return dx
        ops.cln_bwd(dout, x, stats[0], stats[1], t, gw_w, gw_b, dx, g[0], g[1], g[2], g[3], rows, rows_per_sample, C,
                    sample_scale=sample_scale)
        return dx

    def _lean_ok(self, pre, rows, rows_per_sample, C, hid) -> bool:
        """may this ScOTLayer's tail run in the form that keeps no 4C-wide tensor (see `lean_tail`)?  Needs the fused tail in both
        directions, the transposed 16-bit copy of W2, and the gradient arena's contiguous [W1 | b1 | W2 | b2] and norm layouts."""
        key = (pre, rows, rows_per_sample)
        ok = self._lean_cache.get(key)
        if ok is None:
            w2 = pre + ".output.dense.weight"
            ok = (C != 48 and self.use_fused("mlp_bwd", C, rows) and self.use_fused("proj_bwd", C, rows) and self.use_fused("mlp_fwd", C, rows) and self.use_fused("proj_fwd", C, rows)
                  and self.fused_tail and hid == 4 * C and hid % 128 == 0 and rows_per_sample % 64 == 0
                  and self.WT(w2, self.W(w2)) is not None and ops.tail_workgroups(rows, rows_per_sample, C) > 0)
            if ok:
                gs = [self.arena.gview(pre + n) for n in (".intermediate.dense.weight", ".intermediate.dense.bias", ".output.dense.weight", ".output.dense.bias")]
                ok = all(b.data_ptr() == a.data_ptr() + 4 * a.numel() for a, b in zip(gs, gs[1:]))
                ok = ok and self._norm_grads_contiguous(pre + ".layernorm_before", C) and self._norm_grads_contiguous(pre + ".layernorm_after", C)
            self._lean_cache[key] = ok
        return ok

    def _norm_grads_contiguous(self, prefix, C) -> bool:
        """the norm's parameter gradients back to back in the gradient arena (what scot_cln_bwd_finish adds its column sums into)"""
        g = self._norm_grads(prefix)
        ptrs = [t.data_ptr() for t in g if t is not None]
        return all(b - a == 4 * ((C + 63) // 64 * 64) for a, b in zip(ptrs, ptrs[1:]))      # (every arena tensor starts on a 64-float boundary)

    def drop_path_scale(self, prefix: str, B: int, which: int):
        """Swinv2DropPath (HF:565-586): per-sample keep mask / keep_prob for one residual branch of one layer, or None
        (rate 0 / not training).  `drop_path_masks[(prefix, which)]`, when set (tests), overrides the random draw."""
        if self.drop_path_masks is not None:
            m = self.drop_path_masks.get((prefix, which))
            return None if m is None else m.to(self.device, torch.float32)
        rate = self.drop_rates.get(prefix, 0.0)
        if rate <= 0.0:
            return None
        keep = 1.0 - rate
        m = self.new(B)
        self.tdo(lambda: m.bernoulli_(keep).div_(keep))
        return m

    def off_critical_path(self, fn, *tensors):
        """Run fn() (kernel launches that only WRITE parameter gradients) on the side stream, ordered after everything
        enqueued so far on the current stream.  `tensors` are kept alive until the join so the allocator cannot recycle
        them.  The launches are queued and handed over by flush_side(): ONE fork per block instead of one per weight gradient."""
        if not self.use_side or self._in_side is not None:     # no side stream, or already running on it
            fn()
            return
        self._keep.append(tensors)
        self._pending.append(fn)

    def side_stream(self):
        if self.side is None:
            # a stream that is MEASURED to run beside the current one (streams.py: torch's pooled streams and the runtime's hardware queues
            # are both handed out round-robin, and two that share a queue serialise silently)
            from .streams import independent_stream
            self.side = independent_stream(self.device, [torch.cuda.current_stream(self.device)])
        return self.side

    def _run_side(self, fns):
        self.side_stream()
        ev, cur, side = self._event(), torch.cuda.current_stream(), self.side
        self.h_record(ev, cur)
        self.h_wait(side, ev)
        prev = ops.set_workspace_slot(1)
        try:
            with torch.cuda.stream(side):
                for fn in fns:
                    fn()
        finally:
            ops.set_workspace_slot(prev)

    def fork_task(self, fn):
        """Run fn() on the side stream NOW, ordered after everything enqueued so far on the current stream; returns (fn's
        result, event recorded on the side stream after it).  The caller makes the main stream wait for the event
        (`wait_task`) before it consumes what fn produced.  Without a side stream: fn() in line, event None."""
        if not self.use_side:
            return fn(), None
        self.flush_side()
        box = []
        main = torch.cuda.current_stream()

        def run():
            self._in_side = main
            try:
                box.append(fn())
                if self._wgq:
                    self._drain_wgrads()     # the task's own weight gradients: same stream, right behind it
            finally:
                self._in_side = None
        self._task_keep = []
        self._run_side([run])
        ev, side = self._event(), self.side
        self.h_record(ev, side)
        # temporaries of the task come from the main stream's pool: returning them before the main stream is ordered behind the
        # task would hand memory that side-stream kernels still use to the next main-stream allocation
        self._task_keeps[ev], self._task_keep = self._task_keep, []
        return box[0], ev

    def wait_task(self, ev):
        if ev is not None:
            cur = torch.cuda.current_stream()
            self.h_wait(cur, ev)
            self._task_keeps.pop(ev, None)

    def finish_partials(self):
        """The queued partial-sum matrices of norm backwards (one row per workgroup) -> column sums into the parameter gradients, up to
        32 matrices per launch, on the weight-gradient stream.  Called at stage boundaries: one launch for a stage's 16 norms."""
        q, self._finq = self._finq, []
        for i in range(0, len(q), 32):
            part = q[i:i + 32]
            self.off_critical_path(lambda part=part: ops.partial_colsum_batch(part), *[p[0] for p in part])

    def flush_side(self):
        if self._wgq:
            self._drain_wgrads()
        if self._pending:
            fns, self._pending = self._pending, []
            self._run_side(fns)

    def join_side(self):
        self.finish_partials()
        self.flush_side()
        if self.use_side and self.side is not None:
            cur, side = torch.cuda.current_stream(), self.side
            ev = self._event()
            self.h_record(ev, side)
            self.h_wait(cur, ev)
            self._keep.clear()
            self._task_keeps.clear()

    def blocks_fwd(self, blocks, x, x16, B, time, train):
        """The ScOTLayers of one stage → (x, x16, recs); recs is the list of per-block records."""
        recs, q = [], None
        for i, blk in enumerate(blocks):
            last = i + 1 == len(blocks)
            x, x16, r, q = self.layer_fwd(blk, x, x16, B, time, train, qkv_pre=q, next_blk=None if last else blocks[i + 1],
                                          want_attn=self.collect_attn and last, idx=i)      # (a stage returns its LAST block's, ref:859-860)
            recs.append(r)
        return x, x16, recs

    def blocks_bwd(self, recs, g, B, time, split=0, on_upper_half=None):
        """split / on_upper_half: called once the backward has passed block `split` (the gradients of blocks split.. are final: dp.stage_split)"""
        pend = None      # (d_qkv, Wqkv) of the layer just processed, to be applied by the next layer's fused tail as a prologue
        n = len(recs)
        for i, blk_rec in enumerate(reversed(recs)):
            g, pend = self.layer_bwd(blk_rec, g, B, time, pend, defer_qkv_dgrad=(i + 1 < n))
            if split and on_upper_half is not None and n - 1 - i == split:
                on_upper_half()
        assert pend is None
        self.finish_partials()
        self.flush_side()
        return g

    def use_fused(self, part: str, C: int, rows: Optional[int] = None) -> bool:
        """csrc/mlp_fused.hip covers C = 96 / 192 in the 16-bit modes; a tail workgroup owns 64 (128) rows, so below `fused_min_rows` rows the
        launch leaves most CUs idle and the layer-by-layer GEMMs (hundreds of 64 x 64 tiles) win"""
        if C == 48:     # the whole tail as ONE launch per direction (there are no stand-alone C = 48 halves)
            return (self.fused_mlp and self.fused_tail and self.options["fused_fwd48" if part.endswith("_fwd") else "fused_bwd48"]
                    and (rows is None or rows >= self.fused_min_rows))
        return self.fused_mlp and C in (96, 192) and (rows is None or rows >= self.fused_min_rows)

    def wgrad(self, cm, dy, x, gw, b_gelu=False, dbias=None):
        """dW += dy^T x (+ dbias): queued until the next flush_side(), where the queued problems that share a compute mode and
        a token count — the four Linear layers of a ScOTLayer — go out as ONE grouped launch (ops.wgrad_group)."""
        if b_gelu or not self.group_wgrads:
            self.off_critical_path(lambda: ops.linear_wgrad(cm, dy, x, gw, b_gelu=b_gelu, dbias=dbias), dy, x)
            return
        self._wgq.append((cm, dy, x, gw, dbias))

    def _drain_wgrads(self):
        q, self._wgq = self._wgq, []
        groups = {}
        for item in q:
            cm, dy, x, gw, dbias = item
            key = (cm, dy.numel() // dy.shape[-1], dy.dtype, x.dtype)
            groups.setdefault(key, []).append(item)
        for (cm, K, _, _), items in groups.items():
            for i in range(0, len(items), 8):
                part = items[i:i + 8]

                modes = [self.grad_mode(it[3]) for it in part]
                special = any(m != ops.GRAD_ADD for m in modes)

                def run(cm=cm, part=part, modes=modes, special=special):
                    if cm == ops.BF16 and (len(part) > 1 or special) and \
                            ops.wgrad_group(cm, [(dy, x, gw, db) for _, dy, x, gw, db in part], modes if special else None,
                                            self.grad_unscale() if special else None):
                        return
                    for (_, dy, x, gw, db), m in zip(part, modes):
                        self._wgrad_single(cm, dy, x, gw, db, m)
                self.off_critical_path(run, *[t for it in part for t in (it[1], it[2])])

    def _wgrad_single(self, cm, dy, x, gw, db, mode):
        """one weight gradient through scot_gemm (shapes the grouped kernel does not cover), honouring `mode` with plain launches"""
        if mode == ops.GRAD_ADD:
            ops.linear_wgrad(cm, dy, x, gw, dbias=db)
        elif mode == ops.GRAD_STORE_SCALED:
            self.h_zero(gw)
            ops.linear_wgrad(cm, dy, x, gw, dbias=db)
            if self.scale_grads:
                ops.scale_inplace_dev(gw.view(-1), self.grad_unscale())
        else:       # += s · acc through a scratch tensor
            tmp = self.zeros(*gw.shape)
            ops.linear_wgrad(cm, dy, x, tmp, dbias=db)
            ops.axpy_dev(gw.view(-1), tmp.view(-1), self.grad_unscale())

    def linear_bwd_params(self, wname, bname, dy, x, b_gelu=False):
        """dW += dy^T x and db += Σ dy in ONE wgrad launch (the bias sum rides on the dY tiles already in LDS)."""
        self.wgrad(self.compute, dy, x, self.G(wname), b_gelu=b_gelu, dbias=self.G(bname) if bname is not None else None)

    # ------------------------------------------------------------------------------------------ ScOTLayer
    def mark(self, label):
        """SCOT_STAGE_TIMING=1: record a HIP event at a stage boundary (tools/stage_timing.py turns them into a table)."""
        if self.stage_timing:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.marks.append((label, ev))

    def qkv_fusable(self, blk: BlockGeom):
        """May the PREVIOUS layer's fused tail produce this layer's q/k/v projection?  Same channel count by construction (only
        called within a stage); needs the unpadded window geometry (the projection input is then exactly the previous output rows)."""
        H, W = blk.res
        ws, _ = blk.window_shift()
        return H % ws == 0 and W % ws == 0 and blk.dim in self.fused_next_qkv and not self.precision_probe

    def layer_fwd(self, blk: BlockGeom, x, x16, B, time, train, qkv_pre=None, next_blk=None, want_attn=False, idx=0):
        """reference ScOTLayer.forward (model.py:500-581) + Swinv2Attention/Intermediate/Output (HF:389-561).  qkv_pre: this layer's
        q/k/v projection, already produced by the previous layer's fused tail; next_blk: the layer that follows in the same stage
        (its projection becomes this tail's epilogue when the fused tail runs).  Returns (out, out16, rec, qkv_next or None)."""
        cfg, cm = self.cfg, self.compute
        H, W = blk.res
        C, heads = blk.dim, blk.heads
        ws, shift = blk.window_shift()
        Hp, Wp = (H + ws - 1) // ws * ws, (W + ws - 1) // ws * ws
        padded = (Hp, Wp) != (H, W)
        L, Lp = H * W, Hp * Wp
        pre = blk.prefix
        a = pre + ".attention.self."
        # buffers: `tmp` = read only inside this layer and not by the backward (everything, in inference); `nxt` = read by the next layer of
        # the stage only (two alternate: this layer reads the other one).  A stage's last layer writes fresh rows (skips, merges, hidden states).
        par = idx & 1
        tmp = self.pool if not train else (lambda tag, *s_, dtype=torch.float32: self.new(*s_, dtype=dtype))
        alias32 = train and self.adt == torch.float32      # fp32 operands: h16 IS h and out16 IS out, and training keeps the 16-bit rows
        dead = self.pool if not alias32 else tmp           # dead in training too
        fresh_out = next_blk is None or alias32
        nxt = (lambda tag, *s_, dtype=torch.float32: self.pool((tag, par), *s_, dtype=dtype)) if not fresh_out else \
            (lambda tag, *s_, dtype=torch.float32: self.new(*s_, dtype=dtype))
        nxt16 = nxt if not train else (lambda tag, *s_, dtype=torch.float32: self.new(*s_, dtype=dtype))    # (training keeps out16: the next layer's xp)
        if padded:
            xp = tmp("xp", B * Lp, C, dtype=self.adt)
            ops.copy2d(x16, xp, B, H, W, Hp, Wp, C)
        else:
            xp = x16
        wqkv = self.Wspan(a + "qkv_weight", 3 * C * C).view(3 * C, C)
        bqkv = self.arena.span(a + "qkv_bias", 3 * C) if cfg.qkv_bias else None
        ex = self.precision_probe if (self.precision_probe and not train and not padded) else None
        if ex:   # tools/probes/bf16_error_sources.py: selected pieces of an inference forward in fp32 (never on the product path)
            return self._layer_fwd_probe(blk, x, x16, B, time, ex) + (None,)
        if qkv_pre is not None:
            assert not padded
            qkv = qkv_pre
        else:
            qkv = tmp("qkv", B * Lp, 3 * C, dtype=self.adt)
            ops.linear_fwd(cm, xp, wqkv, qkv, bias=bqkv)
        tw = blk.table_window
        if tw != ws:
            raise NotImplementedError("run-time window differs from the constructor-time CPB table window "
                                      f"({ws} vs {tw}); the reference would fail to broadcast here too")
        table = self.cpb_table(pre)   # computed for all layers by the batched launch at the start of forward()
        attn = tmp("attn", B * Lp, C, dtype=self.adt)
        nW = (Hp // ws) * (Wp // ws)
        lse = tmp("lse", B * nW, heads, ws * ws)
        ops.window_attn_fwd(self.acm, qkv, attn, lse, table, self.P(a + "logit_scale"), B, Hp, Wp, C, heads, ws, shift)
        if want_attn:     # output_attentions: the probabilities are recomputed from qkv and lse by a separate kernel, into fresh memory
            probs = torch.empty(B * nW, heads, ws * ws, ws * ws, dtype=torch.float32, device=self.device)
            ops.window_attn_probs(qkv, lse, table, self.P(a + "logit_scale"), probs, B, Hp, Wp, C, heads, ws, shift)
            self.attn_sink.append(probs)
        if padded:
            attn_c = tmp("attn_c", B * L, C, dtype=self.adt)
            ops.copy2d(attn, attn_c, B, Hp, Wp, H, W, C)
        else:
            attn_c = attn
        dp1 = self.drop_path_scale(pre, B, 0) if self.stochastic else None
        dp2 = self.drop_path_scale(pre, B, 1) if self.stochastic else None
        hid = int(cfg.mlp_ratio * C)
        proj_f = self.use_fused("proj_fwd", C, B * L)
        mlp_f = self.use_fused("mlp_fwd", C, B * L) and (hid % 128 == 0 if C != 48 else hid == 192)
        if C == 48 and not (proj_f and mlp_f):
            proj_f = mlp_f = False           # C = 48 exists as the whole tail only
        done_tail = False
        lean_used = False
        qkv_next = None
        if proj_f and mlp_f and self.fused_tail:
            # projection + norm + residual, then MLP + norm + residual, for the same rows in one launch
            lean = train and self.lean_tail and self._lean_ok(pre, B * L, L, C, hid)
            zdt = self.adt if lean else torch.float32        # pre-norm rows: only the norm backward's x-hat reads them
            proj = self.new(B * L, C, dtype=zdt) if train else None
            st1 = (self.new(B * L), self.new(B * L)) if train else (None, None)
            h, h16 = dead("h", B * L, C), tmp("h16", B * L, C, dtype=self.adt)
            u = self.new(B * L, hid, dtype=self.adt) if (train and not lean) else None
            gp = self.new(B * L, hid, dtype=self.adt) if train else None
            y2 = self.new(B * L, C, dtype=zdt) if train else None
            st2 = (self.new(B * L), self.new(B * L)) if train else (None, None)
            out, out16 = nxt("out", B * L, C), nxt16("out16", B * L, C, dtype=self.adt)
            n1, n2 = self._norm_params(pre + ".layernorm_before"), self._norm_params(pre + ".layernorm_after")
            nq = (None, None, None)
            if next_blk is not None and next_blk.dim == C and self.qkv_fusable(next_blk):
                na = next_blk.prefix + ".attention.self."
                qkv_next = nxt16("qkvn", B * L, 3 * C, dtype=self.adt)
                nq = (self.Wspan(na + "qkv_weight", 3 * C * C).view(3 * C, C),
                      self.arena.span(na + "qkv_bias", 3 * C) if cfg.qkv_bias else None, qkv_next)
            done_tail = ops.block_tail_fwd(
                (attn_c, self.W(pre + ".attention.output.dense.weight"), self.P(pre + ".attention.output.dense.bias"), x, h, h16, proj,
                 st1[0], st1[1], n1[0], n1[1], n1[2], n1[3], dp1),
                (self.W(pre + ".intermediate.dense.weight"), self.P(pre + ".intermediate.dense.bias"), self.W(pre + ".output.dense.weight"),
                 self.P(pre + ".output.dense.bias"), out, out16, u, gp, y2, st2[0], st2[1], n2[0], n2[1], n2[2], n2[3], dp2),
                time if self.cond else None, B * L, L, C, hid, cfg.layer_norm_eps, *nq, z16=lean)
            lean_used = lean
            if not done_tail:
                if lean:
                    raise RuntimeError("scot_block_tail_fwd rejected a shape the engine selected it for")
                qkv_next = None
                if C == 48:
                    proj_f = mlp_f = False   # layer by layer
        if done_tail:
            pass
        elif proj_f:
            proj = self.new(B * L, C) if train else None
            st1 = (self.new(B * L), self.new(B * L)) if train else (None, None)
            h, h16 = dead("h", B * L, C), tmp("h16", B * L, C, dtype=self.adt)
            gw_w, gw_b, bw_w, bw_b = self._norm_params(pre + ".layernorm_before")
            if not ops.proj_cln_fwd(attn_c, self.W(pre + ".attention.output.dense.weight"), self.P(pre + ".attention.output.dense.bias"),
                                    x, h, h16, proj, st1[0], st1[1], time if self.cond else None, gw_w, gw_b, bw_w, bw_b, dp1, B * L, L,
                                    C, cfg.layer_norm_eps):
                raise RuntimeError("scot_proj_cln_fwd rejected a shape the engine selected it for")
        else:
            proj = tmp("proj", B * L, C)
            ops.linear_fwd(cm, attn_c, self.W(pre + ".attention.output.dense.weight"), proj,
                           bias=self.P(pre + ".attention.output.dense.bias"))
            h, h16, st1 = self.norm_fwd(pre + ".layernorm_before", proj, x, L, C, cfg.layer_norm_eps, time, need_stats=train,
                                        copy=True, sample_scale=dp1, out=dead("h", B * L, C), out16=tmp("h16", B * L, C, dtype=self.adt))
        if done_tail:
            pass
        elif mlp_f:
            u = self.new(B * L, hid, dtype=self.adt) if train else None
            gp = self.new(B * L, hid, dtype=self.adt) if train else None
            y2 = self.new(B * L, C) if train else None
            st2 = (self.new(B * L), self.new(B * L)) if train else (None, None)
            out, out16 = nxt("out", B * L, C), nxt16("out16", B * L, C, dtype=self.adt)
            gw_w, gw_b, bw_w, bw_b = self._norm_params(pre + ".layernorm_after")
            if not ops.mlp_block_fwd(h16, h, self.W(pre + ".intermediate.dense.weight"), self.P(pre + ".intermediate.dense.bias"),
                                     self.W(pre + ".output.dense.weight"), self.P(pre + ".output.dense.bias"), out, out16, u, gp, y2,
                                     st2[0], st2[1], time if self.cond else None, gw_w, gw_b, bw_w, bw_b, dp2, B * L, L, C, hid,
                                     cfg.layer_norm_eps):
                raise RuntimeError("scot_mlp_block_fwd rejected a shape the engine selected it for")
        else:
            # fc1 epilogue emits a = gelu(u) AND gp = gelu'(u) (one erf, fp32 registers); u itself is never stored
            u = tmp("u", B * L, hid, dtype=self.adt)
            gp = self.new(B * L, hid, dtype=self.adt) if train else None
            ops.linear_fwd(cm, h16, self.W(pre + ".intermediate.dense.weight"), u, bias=self.P(pre + ".intermediate.dense.bias"),
                           gelu_deriv_out=gp if train else u)     # eval: GELU(u) only (`gelu_deriv_out is out`)
            y2 = tmp("y2", B * L, C)
            ops.linear_fwd(cm, u, self.W(pre + ".output.dense.weight"), y2, bias=self.P(pre + ".output.dense.bias"))
            out, out16, st2 = self.norm_fwd(pre + ".layernorm_after", y2, h, L, C, cfg.layer_norm_eps, time, need_stats=train,
                                            copy=True, sample_scale=dp2, out=nxt("out", B * L, C), out16=nxt16("out16", B * L, C, dtype=self.adt))
        rec = None
        if train:
            rec = dict(blk=blk, xp=xp, qkv=qkv, attn_p=attn, table=table, lse=lse, attn_c=attn_c, proj=proj, st1=st1, h16=h16, u=u, gp=gp,
                       y2=y2, st2=st2, geom=(H, W, Hp, Wp, ws, shift, padded), dp=(dp1, dp2), lean=bool(done_tail and lean_used))
        return out, out16, rec, qkv_next

    def dgrad_into(self, cm, dy, w, g, wt=None):
        """g += dy·w, in place (every reader of g launched so far is on the same stream)"""
        ops.linear_dgrad(cm, dy, w, g, accumulate=True, wt=wt)
        return g

    def _layer_fwd_probe(self, blk, x, x16, B, time, ex):
        """Inference forward of one ScOTLayer with chosen pieces in fp32 (ex: set of 'qkv', 'attn', 'proj', 'mlp'): measures
        where the bf16 mode's error comes from.  Not used by any product path."""
        cfg, cm = self.cfg, self.compute
        H, W = blk.res
        C, heads, pre = blk.dim, blk.heads, blk.prefix
        ws, shift = blk.window_shift()
        L = H * W
        a = pre + ".attention.self."
        f32 = torch.float32
        bqkv = self.arena.span(a + "qkv_bias", 3 * C) if cfg.qkv_bias else None
        qdt = f32 if ("attn" in ex or "qkv" in ex) else self.adt
        qkv = self.new(B * L, 3 * C, dtype=qdt)
        if "qkv" in ex:
            ops.linear_fwd(ops.F32, x, self.arena.span(a + "qkv_weight", 3 * C * C).view(3 * C, C), qkv, bias=bqkv)
        else:
            ops.linear_fwd(cm, x16, self.Wspan(a + "qkv_weight", 3 * C * C).view(3 * C, C), qkv, bias=bqkv)
        table = self.cpb_table(pre)
        nW = (H // ws) * (W // ws)
        lse = self.new(B * nW, heads, ws * ws)
        if "attn" in ex:
            if qkv.dtype != f32:
                q2 = self.new(B * L, 3 * C)
                ops.cast(qkv, q2)
                qkv = q2
            attn = self.new(B * L, C)
            ops.window_attn_fwd(ops.F32, qkv, attn, lse, table, self.P(a + "logit_scale"), B, H, W, C, heads, ws, shift)
        else:
            if qkv.dtype != self.adt:
                q2 = self.new(B * L, 3 * C, dtype=self.adt)
                ops.cast(qkv, q2)
                qkv = q2
            attn = self.new(B * L, C, dtype=self.adt)
            ops.window_attn_fwd(cm, qkv, attn, lse, table, self.P(a + "logit_scale"), B, H, W, C, heads, ws, shift)
        proj = self.new(B * L, C)
        if "proj" in ex:
            if attn.dtype != f32:
                a2 = self.new(B * L, C)
                ops.cast(attn, a2)
                attn = a2
            ops.linear_fwd(ops.F32, attn, self.arena.view(pre + ".attention.output.dense.weight"), proj,
                           bias=self.P(pre + ".attention.output.dense.bias"))
        else:
            if attn.dtype != self.adt:
                a2 = self.new(B * L, C, dtype=self.adt)
                ops.cast(attn, a2)
                attn = a2
            ops.linear_fwd(cm, attn, self.W(pre + ".attention.output.dense.weight"), proj, bias=self.P(pre + ".attention.output.dense.bias"))
        h, h16, _ = self.norm_fwd(pre + ".layernorm_before", proj, x, L, C, cfg.layer_norm_eps, time, need_stats=False, copy=True)
        hid = int(cfg.mlp_ratio * C)
        y2 = self.new(B * L, C)
        if "mlp" in ex:
            u = self.new(B * L, hid)
            ops.linear_fwd(ops.F32, h, self.arena.view(pre + ".intermediate.dense.weight"), u, bias=self.P(pre + ".intermediate.dense.bias"),
                           gelu_deriv_out=u)
            ops.linear_fwd(ops.F32, u, self.arena.view(pre + ".output.dense.weight"), y2, bias=self.P(pre + ".output.dense.bias"))
        else:
            u = self.new(B * L, hid, dtype=self.adt)
            ops.linear_fwd(cm, h16, self.W(pre + ".intermediate.dense.weight"), u, bias=self.P(pre + ".intermediate.dense.bias"),
                           gelu_deriv_out=u)
            ops.linear_fwd(cm, u, self.W(pre + ".output.dense.weight"), y2, bias=self.P(pre + ".output.dense.bias"))
        out, out16, _ = self.norm_fwd(pre + ".layernorm_after", y2, h, L, C, cfg.layer_norm_eps, time, need_stats=False, copy=True)
        return out, out16, None

    def layer_bwd(self, rec, g, B, time, pend=None, defer_qkv_dgrad=False):
        """g: fp32 [B*L, C] gradient wrt the layer output; returns (gradient wrt the layer input — the same buffer —,
        pending).  pend = (d_qkv, Wqkv) of the layer processed before this one whose qkv dgrad `g += d_qkv·Wqkv`
        has not been applied yet: the fused block tail does it as a prologue (same rows), otherwise it is launched here first.
        defer_qkv_dgrad: leave THIS layer's qkv dgrad to the next call in the same way when that call can take it."""
        cfg, cm, adt = self.cfg, self.compute, self.adt
        blk: BlockGeom = rec["blk"]
        H, W, Hp, Wp, ws, shift, padded = rec["geom"]
        C, heads, pre = blk.dim, blk.heads, blk.prefix
        a = pre + ".attention.self."
        L, Lp = H * W, Hp * Wp
        hid = int(cfg.mlp_ratio * C)
        mlp_f = self.use_fused("mlp_bwd", C, B * L) and (hid % 128 == 0 if C != 48 else hid == 192) and L % 64 == 0
        proj_f = self.use_fused("proj_bwd", C, B * L) and L % 64 == 0
        if C == 48 and not (mlp_f and proj_f and rec.get("gp") is not None):
            mlp_f = proj_f = False           # C = 48 exists as the whole tail only
        tail_f = mlp_f and proj_f and self.fused_tail
        can_prologue = tail_f and C in self.fused_qkv_dgrad and not padded
        if pend is not None and not can_prologue:
            g = self.dgrad_into(cm, pend[0], pend[1], g, wt=pend[2])
            pend = None
        d_attn = self.pool("d_attn", B * L, C, dtype=adt)      # read by this layer's attention backward (same stream) and by nothing else
        done_tail = False
        lean = bool(rec.get("lean"))
        if lean:
            # the tail without 4C-wide tensors: gelu'(u) recomputed from h16, du never stored, the norms' parameter gradients as
            # per-workgroup partial rows; on the weight-gradient stream: the partial rows' column sums, the fc1 / fc2 gradients with
            # gelu(u) / du recomputed (scot_wgrad_mlp), the out-projection's (and, below, the qkv projection's) through the grouped GEMM
            if not tail_f:
                raise RuntimeError("the forward kept no gelu(u) / gelu'(u) for this layer, but the backward's fused tail is switched off")
            d_y2, d_proj = self.new(B * L, C, dtype=adt), self.new(B * L, C, dtype=adt)
            n2, n1 = self._norm_params(pre + ".layernorm_after"), self._norm_params(pre + ".layernorm_before")
            g2, g1 = self._norm_grads(pre + ".layernorm_after"), self._norm_grads(pre + ".layernorm_before")
            nwg, ncol = ops.tail_workgroups(B * L, L, C), (4 if self.cond else 2) * ((C + 63) // 64 * 64)
            part2, part1 = self.new(nwg, ncol), self.new(nwg, ncol)
            w1n, w2n = pre + ".intermediate.dense.weight", pre + ".output.dense.weight"
            b1 = self.P(pre + ".intermediate.dense.bias")
            if not ops.block_tail_bwd(
                    g, g,
                    (rec["y2"], rec["st2"][0], rec["st2"][1], n2[0], n2[1], rec["dp"][1], rec["gp"], self.W(w1n), self.W(w2n), d_y2, None,
                     None, None, None, None),
                    (rec["proj"], rec["st1"][0], rec["st1"][1], n1[0], n1[1], rec["dp"][0], self.W(pre + ".attention.output.dense.weight"),
                     d_proj, d_attn, None, None, None, None),
                    time if self.cond else None, B * L, L, C, hid, dqkv=pend[0] if pend else None, wqkv=pend[1] if pend else None,
                    h16=rec["h16"], b1=b1, z16=True, partial2=part2, partial1=part1):
                raise RuntimeError("scot_block_tail_bwd rejected a shape the engine selected it for")
            pend = None
            done_tail = True
            first2, first1 = next(t for t in g2 if t is not None), next(t for t in g1 if t is not None)
            h16r, w2t = rec["h16"], self.WT(w2n, self.W(w2n))
            gW1, gb1, gW2, gb2 = self.G(w1n), self.G(pre + ".intermediate.dense.bias"), self.G(w2n), self.G(pre + ".output.dense.bias")

            self._finq += [(part2, nwg, ncol, first2), (part1, nwg, ncol, first1)]

            mlp_mode = self.grad_mode(gW1)
            assert mlp_mode == self.grad_mode(gW2)

            def side():
                if not ops.wgrad_mlp(h16r, d_y2, self.W(w1n), b1, w2t, gW1, gb1, gW2, gb2, mode=mlp_mode,
                                     grad_scale=self.grad_unscale() if mlp_mode != ops.GRAD_ADD else None):
                    raise RuntimeError("scot_wgrad_mlp rejected a shape the engine selected it for")
            self.off_critical_path(side, h16r, d_y2)
            self.linear_bwd_params(pre + ".attention.output.dense.weight", pre + ".attention.output.dense.bias", d_proj, rec["attn_c"])
        elif tail_f:
            # both halves of the block tail in one launch: the residual-stream gradient between them stays in registers
            d_y2, d_u, d_proj = self.new(B * L, C, dtype=adt), self.new(B * L, hid, dtype=adt), self.new(B * L, C, dtype=adt)
            n2, g2 = self._norm_params(pre + ".layernorm_after"), self._norm_grads(pre + ".layernorm_after")
            n1, g1 = self._norm_params(pre + ".layernorm_before"), self._norm_grads(pre + ".layernorm_before")
            gout = g
            done_tail = ops.block_tail_bwd(
                g, gout,
                (rec["y2"], rec["st2"][0], rec["st2"][1], n2[0], n2[1], rec["dp"][1], rec["gp"], self.W(pre + ".intermediate.dense.weight"),
                 self.W(pre + ".output.dense.weight"), d_y2, d_u, g2[0], g2[1], g2[2], g2[3]),
                (rec["proj"], rec["st1"][0], rec["st1"][1], n1[0], n1[1], rec["dp"][0], self.W(pre + ".attention.output.dense.weight"),
                 d_proj, d_attn, g1[0], g1[1], g1[2], g1[3]),
                time if self.cond else None, B * L, L, C, hid, dqkv=pend[0] if pend else None, wqkv=pend[1] if pend else None)
            if not done_tail and pend is not None:
                g = self.dgrad_into(cm, pend[0], pend[1], g, wt=pend[2])
            pend = None
            if not done_tail and C == 48:
                mlp_f = proj_f = False       # layer by layer
            if done_tail:
                g = gout
                self.linear_bwd_params(pre + ".output.dense.weight", pre + ".output.dense.bias", d_y2, rec["u"])
                self.linear_bwd_params(pre + ".intermediate.dense.weight", pre + ".intermediate.dense.bias", d_u, rec["h16"])
                self.linear_bwd_params(pre + ".attention.output.dense.weight", pre + ".attention.output.dense.bias", d_proj, rec["attn_c"])
        if done_tail:
            pass
        elif mlp_f:
            # the whole dependent chain of the MLP half in one launch; the two weight gradients follow on the side stream
            d_y2 = self.new(B * L, C, dtype=adt)
            d_u = self.new(B * L, hid, dtype=adt)
            gw_w, gw_b, _, _ = self._norm_params(pre + ".layernorm_after")
            gg = self._norm_grads(pre + ".layernorm_after")
            g2 = g
            if not ops.mlp_block_bwd(g, g2, rec["y2"], rec["st2"][0], rec["st2"][1], time if self.cond else None, gw_w, gw_b,
                                     rec["dp"][1], rec["gp"], self.W(pre + ".intermediate.dense.weight"),
                                     self.W(pre + ".output.dense.weight"), d_y2, d_u, gg[0], gg[1], gg[2], gg[3], B * L, L, C, hid):
                raise RuntimeError("scot_mlp_block_bwd rejected a shape the engine selected it for")
            g = g2
            self.linear_bwd_params(pre + ".output.dense.weight", pre + ".output.dense.bias", d_y2, rec["u"])
            self.linear_bwd_params(pre + ".intermediate.dense.weight", pre + ".intermediate.dense.bias", d_u, rec["h16"])
        else:
            # out = h + CLN_after(y2)
            d_y2 = self.norm_bwd(pre + ".layernorm_after", g, rec["y2"], rec["st2"], L, C, time, adt, sample_scale=rec["dp"][1])
            # y2 = gelu(u) W2^T + b2
            self.linear_bwd_params(pre + ".output.dense.weight", pre + ".output.dense.bias", d_y2, rec["u"])   # rec["u"] = gelu(u)
            d_u = self.new(B * L, hid, dtype=adt)
            ops.linear_dgrad(cm, d_y2, self.W(pre + ".output.dense.weight"), d_u, aux=rec["gp"], aux_mul=True,
                             wt=self.WT(pre + ".output.dense.weight"))
            # u = h W1^T + b1
            self.linear_bwd_params(pre + ".intermediate.dense.weight", pre + ".intermediate.dense.bias", d_u, rec["h16"])
            g = self.dgrad_into(cm, d_u, self.W(pre + ".intermediate.dense.weight"), g, wt=self.WT(pre + ".intermediate.dense.weight"))
        # h = x + CLN_before(proj)
        if done_tail:
            pass
        elif proj_f:
            d_proj = self.new(B * L, C, dtype=adt)
            gw_w, gw_b, _, _ = self._norm_params(pre + ".layernorm_before")
            gg = self._norm_grads(pre + ".layernorm_before")
            if not ops.proj_cln_bwd(g, rec["proj"], rec["st1"][0], rec["st1"][1], time if self.cond else None, gw_w, gw_b, rec["dp"][0],
                                    self.W(pre + ".attention.output.dense.weight"), d_proj, d_attn, gg[0], gg[1], gg[2], gg[3], B * L,
                                    L, C):
                raise RuntimeError("scot_proj_cln_bwd rejected a shape the engine selected it for")
            self.linear_bwd_params(pre + ".attention.output.dense.weight", pre + ".attention.output.dense.bias", d_proj, rec["attn_c"])
        else:
            d_proj = self.norm_bwd(pre + ".layernorm_before", g, rec["proj"], rec["st1"], L, C, time, adt, sample_scale=rec["dp"][0])
            self.linear_bwd_params(pre + ".attention.output.dense.weight", pre + ".attention.output.dense.bias", d_proj, rec["attn_c"])
            ops.linear_dgrad(cm, d_proj, self.W(pre + ".attention.output.dense.weight"), d_attn,
                             wt=self.WT(pre + ".attention.output.dense.weight"))
        if padded:
            d_attn_p = self.pool("d_attn_p", B * Lp, C, dtype=adt)
            ops.copy2d(d_attn, d_attn_p, B, H, W, Hp, Wp, C)
        else:
            d_attn_p = d_attn
        d_qkv = self.new(B * Lp, 3 * C, dtype=adt)
        d_table = self.cpb_table(pre, grad=True)   # zeroed once per backward; its MLP backward is batched per stage
        if self.attn_rep > 1:
            lo = self.cpb_ls_off[pre]
            ops.window_attn_bwd_rep(self.acm, rec["qkv"], rec["attn_p"], d_attn_p, rec["lse"], rec["table"], self.P(a + "logit_scale"), d_qkv,
                                    d_table, self.cpb_dls[lo:lo + heads], B, Hp, Wp, C, heads, ws, shift, self.attn_rep, self.cpb_tab_total,
                                    self.cpb_ls_total)
        else:
            ops.window_attn_bwd(self.acm, rec["qkv"], rec["attn_p"], d_attn_p, rec["lse"], rec["table"], self.P(a + "logit_scale"), d_qkv,
                                d_table, self.G(a + "logit_scale"), B, Hp, Wp, C, heads, ws, shift)
        wqkv = self.Wspan(a + "qkv_weight", 3 * C * C).view(3 * C, C)
        gwqkv = self.arena.span(a + "qkv_weight", 3 * C * C, grad=True).view(3 * C, C)
        self.wgrad(cm, d_qkv, rec["xp"], gwqkv, dbias=self.arena.span(a + "qkv_bias", 3 * C, grad=True) if cfg.qkv_bias else None)
        if padded:
            tmp = self.new(B * Lp, C)
            ops.linear_dgrad(cm, d_qkv, wqkv, tmp, wt=self.WT(a + "qkv_weight"))
            tmpc = self.new(B * L, C)
            ops.copy2d(tmp, tmpc, B, Hp, Wp, H, W, C)
            g2 = g
            ops.add(g, tmpc, g2)
            g = g2
            new_pend = None
        elif defer_qkv_dgrad and can_prologue:
            new_pend = (d_qkv, wqkv, self.WT(a + "qkv_weight"))      # (the next layer of this stage has the same geometry: its fused tail applies it)
        else:
            g = self.dgrad_into(cm, d_qkv, wqkv, g, wt=self.WT(a + "qkv_weight"))
            new_pend = None
        self.flush_side()
        return g, new_pend

    # ------------------------------------------------------------------------------------------ resampling
    def merge_fwd(self, st: StageGeom, x, stage_in, B, time, train):
        """reference ScOTPatchMerging (model.py:680-712) on (stage_out + stage_in) (model.py:847-849)."""
        H, W = st.res
        C = st.dim
        H2, W2 = (H + 1) // 2, (W + 1) // 2
        cat = self.new(B * H2 * W2, 4 * C, dtype=self.tadt)
        ops.space_to_depth(x, stage_in, cat, B, H, W, C, 0)
        r = self.new(B * H2 * W2, 2 * C)
        ops.linear_fwd(self.tcm, cat, self.TW(st.prefix + ".downsample.reduction.weight"), r)
        out, out16, stats = self.norm_fwd(st.prefix + ".downsample.norm", r, None, H2 * W2, 2 * C, 1e-5, time, need_stats=train, copy=True)
        return out, out16, (dict(cat=cat, r=r, stats=stats) if train else None)

    def merge_bwd(self, st: StageGeom, rec, g, B, time):
        H, W = st.res
        C = st.dim
        H2, W2 = (H + 1) // 2, (W + 1) // 2
        d_r = self.norm_bwd(st.prefix + ".downsample.norm", g, rec["r"], rec["stats"], H2 * W2, 2 * C, time, self.tadt)
        self.wgrad(self.tcm, d_r, rec["cat"], self.G(st.prefix + ".downsample.reduction.weight"))
        d_cat = self.new(B * H2 * W2, 4 * C)
        ops.linear_dgrad(self.tcm, d_r, self.TW(st.prefix + ".downsample.reduction.weight"), d_cat,
                         wt=self.WT(st.prefix + ".downsample.reduction.weight", self.TW(st.prefix + ".downsample.reduction.weight")))
        d_sum = self.new(B * H * W, C)
        ops.depth_to_space(d_cat, d_sum, B, H, W, H2, W2, C, 0)
        return d_sum

    def unmerge_fwd(self, st: StageGeom, x, x16, B, time, train):
        """reference ScOTPatchUnmerging (model.py:737-760)."""
        h, w = st.res
        oh, ow = st.out_res
        C = st.dim
        xin = x if self.tadt == torch.float32 else x16
        up = self.new(B * h * w, 2 * C, dtype=self.tadt)
        ops.linear_fwd(self.tcm, xin, self.TW(st.prefix + ".upsample.upsample.weight"), up)
        sh = self.new(B * oh * ow, C // 2, dtype=self.tadt)
        ops.depth_to_space(up, sh, B, oh, ow, h, w, C // 2, 1)
        n, _, stats = self.norm_fwd(st.prefix + ".upsample.norm", sh, None, oh * ow, C // 2, 1e-5, time, out_dtype=self.tadt,
                                    need_stats=train)
        out = self.new(B * oh * ow, C // 2)
        ops.linear_fwd(self.tcm, n, self.TW(st.prefix + ".upsample.mixup.weight"), out)
        return out, self.to_adt(out), (dict(x=xin, sh=sh, stats=stats, n=n) if train else None)

    def unmerge_bwd(self, st: StageGeom, rec, g, B, time):
        h, w = st.res
        oh, ow = st.out_res
        C = st.dim
        g16 = self.to_tadt(g)
        self.wgrad(self.tcm, g16, rec["n"], self.G(st.prefix + ".upsample.mixup.weight"))
        d_n = self.new(B * oh * ow, C // 2, dtype=self.tadt)
        ops.linear_dgrad(self.tcm, g16, self.TW(st.prefix + ".upsample.mixup.weight"), d_n,
                         wt=self.WT(st.prefix + ".upsample.mixup.weight", self.TW(st.prefix + ".upsample.mixup.weight")))
        d_sh = self.norm_bwd(st.prefix + ".upsample.norm", d_n, rec["sh"], rec["stats"], oh * ow, C // 2, time, self.tadt)
        d_up = self.new(B * h * w, 2 * C, dtype=self.tadt)
        ops.space_to_depth(d_sh, None, d_up, B, oh, ow, C // 2, 1)
        self.wgrad(self.tcm, d_up, rec["x"], self.G(st.prefix + ".upsample.upsample.weight"))
        gx = self.new(B * h * w, C)
        ops.linear_dgrad(self.tcm, d_up, self.TW(st.prefix + ".upsample.upsample.weight"), gx,
                         wt=self.WT(st.prefix + ".upsample.upsample.weight", self.TW(st.prefix + ".upsample.upsample.weight")))
        return gx

    # ------------------------------------------------------------------------------------------ ConvNeXt skip block
    def convnext_fwd(self, pre, s, B, H, W, C, time, train):
        """reference ConvNeXtBlock.forward (model.py:198-217)."""
        L = H * W
        dw = self.new(B * L, C)
        ops.dwconv7(s, self.P(pre + ".dwconv.weight"), self.P(pre + ".dwconv.bias"), dw, B, H, W, C)
        n, _, stats = self.norm_fwd(pre + ".norm", dw, None, L, C, self.cfg.layer_norm_eps, time, out_dtype=self.adt, need_stats=train)
        u = self.new(B * L, 4 * C, dtype=self.adt)
        gp = self.new(B * L, 4 * C, dtype=self.adt) if train else None
        # the epilogue stores GELU(v) (and GELU'(v) when training; `gelu_deriv_out is out` = value only)
        ops.linear_fwd(self.compute, n, self.W(pre + ".pwconv1.weight"), u, bias=self.P(pre + ".pwconv1.bias"),
                       gelu_deriv_out=gp if train else u)
        y2 = self.new(B * L, C)
        ops.linear_fwd(self.compute, u, self.W(pre + ".pwconv2.weight"), y2, bias=self.P(pre + ".pwconv2.bias"))
        out = self.new(B * L, C)
        ops.scale_residual(y2, self.P(pre + ".weight"), s, out, B * L, C)
        return out, (dict(s=s, dw=dw, stats=stats, n=n, u=u, gp=gp, y2=y2) if train else None)

    def convnext_bwd(self, pre, rec, g, B, H, W, C, time):
        L = H * W
        ls = self._ls.get(pre) if C % 8 == 0 else None
        d_y2 = self.new(B * L, C, dtype=self.adt)
        if ls is not None:
            # binary16 operands: the branch's gradients are (g ⊙ γ)·c with c = 2^k bringing max|γ| into (1/2, 1]; its parameter
            # gradients go to the block's scratch range and are handed to the arena divided by c at the end (all on the device)
            ops.pow2_rescale(self.P(pre + ".weight"), ls["cs"])
            ops.colscale_dev(g, self.P(pre + ".weight"), ls["cs"][0:1], d_y2, B * L, C)
            self._gredirect = ls["views"]
        else:
            ops.scale_residual(g, self.P(pre + ".weight"), None, d_y2, B * L, C)
        try:
            return self._convnext_bwd_chain(pre, rec, g, d_y2, B, H, W, C, time, ls)
        finally:
            self._gredirect = None

    def _convnext_bwd_chain(self, pre, rec, g, d_y2, B, H, W, C, time, ls):
        L = H * W
        # layer-scale gradient Σ g·y2: reads g BEFORE the in-place `g += d_s` at the end of this function — keep it on this stream
        ops.colsum(g, self.arena.gview(pre + ".weight"), y=rec["y2"])
        self.linear_bwd_params(pre + ".pwconv2.weight", pre + ".pwconv2.bias", d_y2, rec["u"])
        d_u = self.new(B * L, 4 * C, dtype=self.adt)
        ops.linear_dgrad(self.compute, d_y2, self.W(pre + ".pwconv2.weight"), d_u, aux=rec["gp"], aux_mul=True, wt=self.WT(pre + ".pwconv2.weight"))
        self.linear_bwd_params(pre + ".pwconv1.weight", pre + ".pwconv1.bias", d_u, rec["n"])
        d_n = self.new(B * L, C, dtype=self.adt)
        ops.linear_dgrad(self.compute, d_u, self.W(pre + ".pwconv1.weight"), d_n, wt=self.WT(pre + ".pwconv1.weight"))
        d_dw = self.norm_bwd(pre + ".norm", d_n, rec["dw"], rec["stats"], L, C, time, torch.float32)
        self.off_critical_path(lambda: ops.dwconv7_wgrad(d_dw, rec["s"], self.G(pre + ".dwconv.weight"), self.G(pre + ".dwconv.bias"),
                                                         B, H, W, C), d_dw, rec["s"])
        d_s = self.new(B * L, C)
        ops.dwconv7(d_dw, self.P(pre + ".dwconv.weight"), None, d_s, B, H, W, C, flip=True)
        if ls is None:
            ops.add(g, d_s, g)
            return g
        ops.axpy_dev(g, d_s, ls["cs"][1:2])
        # scratch / c -> the arena, behind every kernel that accumulated into the scratch (weight gradients, the norm's partial sums)
        self.finish_partials()
        self._drain_wgrads()
        dst = self.arena.grad[ls["lo"]:ls["hi"]]
        self.off_critical_path(lambda: ops.axpy_dev(dst, ls["scratch"], ls["cs"][1:2], clear_src=True))
        return g

    # ------------------------------------------------------------------------------------------ whole model
    def refresh_weight_copies(self, train: bool):
        """fp32 master weights -> 16-bit GEMM operands (+ the transposed copies the data gradients read), when the master changed
        since the copies were made.  Never part of a recorded step: whether it runs is decided per call."""
        if self.shadow is None:
            return
        v = self.weights_version() if self.weights_version is not None else None
        need = v is None or v != self._shadow_v
        need_t = train and self.shadow_t is not None and (v is None or v != self._shadow_t_v)
        if not (need or need_t):
            return
        prev = ops.set_recorder(None)
        try:
            if need:
                ops.cast(self.arena.data, self.shadow)
                self._shadow_v = v
            if need_t:
                self.transpose_weights()
                self._shadow_t_v = v
        finally:
            ops.set_recorder(prev)

    def weight_copies_are_current(self, v):
        """the optimizer has just written both copies from master weights whose version is `v`"""
        self._copies_maintained = True
        self._shadow_v = v
        if self.shadow_t is not None:
            self._shadow_t_v = v

    def forward(self, pixel_values, time=None, labels=None, pixel_mask=None, train=True, stochastic=None, bool_masked_pos=None):
        prev = ops.use(self.lib_kind)
        try:
            if not self._capturing():
                self.refresh_weight_copies(train)
            elif self.shadow is not None and not self._copies_maintained:
                # a hipGraph is being captured and nobody but this engine keeps the 16-bit copies current (no FusedAdamW): the cast
                # and the transposes become part of the captured step, as they were before the version gate existed — a replay after a
                # foreign optimizer's update would otherwise multiply by stale weights for ever
                prev_rec = ops.set_recorder(None)
                try:
                    ops.cast(self.arena.data, self.shadow)
                    if train and self.shadow_t is not None:
                        self.transpose_weights()
                finally:
                    ops.set_recorder(prev_rec)
            if bool_masked_pos is not None:       # masked-position pre-training inputs: not a taped signature
                self.stochastic = bool(train if stochastic is None else stochastic)
                return self._forward(pixel_values, time, labels, pixel_mask, train, bool_masked_pos)
            return self._forward_step(pixel_values, time, labels, pixel_mask, train, stochastic)
        finally:
            ops.use(prev)

    def backward(self, tape, dloss=None, dpred=None):
        prev = ops.use(self.lib_kind)
        try:
            return self._backward_step(tape, dloss, dpred)
        finally:
            ops.use(prev)
            self.grads_are_zero = False
            self.lazy_grads = False

    def _forward_step(self, pixel_values, time=None, labels=None, pixel_mask=None, train=True, stochastic=None):
        """→ (loss [1] or None, prediction [B,Cout,H,W], tape or None).  Inputs: fp32 contiguous CUDA tensors.
        `train` = keep what the backward needs; `stochastic` = draw stochastic-depth masks (the reference keys that on
        `module.training`, HF:565-586; default: same as `train`).

        Training steps go through the step tape (see __init__): call 1 of a signature runs the ops directly, call 2 runs them
        and records, later calls copy the inputs into the recorded step's input buffers and replay.  A recorded step has ONE
        set of activation buffers, so it is replayed only while no earlier forward of it still waits for its backward
        (forward-forward-backward-backward, the reference's AR training loop trainer.py:466-490, takes the untaped path for
        the second forward); loss and prediction are returned as fresh tensors, never as views of the recorded buffers."""
        self.stochastic = bool(train if stochastic is None else stochastic)
        self._events.clear()          # (the previous step's: a destroyed event's pending work completes regardless)
        if (not self.tape_mode or (train and labels is None) or self.stage_timing or self.collect_attn or self._capturing()
                or (not train and (self.stochastic or not self.tape_inference))):
            return self._forward(pixel_values, time, labels, pixel_mask, train)
        # (inference forwards are taped too — an autoregressive rollout is hundreds of forwards of one signature, and issued through the
        # Python op wrappers a forward is host-bound: 400 launches at ~10 us each against ~5.5 ms of GPU time)
        key = (tuple(pixel_values.shape), None if time is None else tuple(time.shape), None if labels is None else tuple(labels.shape),
               None if pixel_mask is None else (tuple(pixel_mask.shape), pixel_mask.dtype), self._stream_id(), self.stochastic, bool(train))
        ent = self._taped.get(key)
        if ent is None:
            # a recorded step pins all of its buffers (GBs): keep at most `tape_max` signatures (e.g. the full batch and the
            # epoch's short last batch); the least recently used one is dropped, its buffers go back to the allocator
            # Inference signatures (small: no backward state) are counted on their own, so an eval pass between epochs never evicts the
            # training tapes (several GB each, and possibly a recorded forward still waiting for its backward)
            same = [k for k in self._taped if k[-1] == key[-1]]
            while len(same) >= self.tape_max:
                self._taped.pop(same.pop(0))
            self._taped[key] = dict(state="warm")
            return self._forward(pixel_values, time, labels, pixel_mask, train)
        self._taped[key] = self._taped.pop(key)   # most recently used last
        ins = (pixel_values, time, labels, pixel_mask)
        pend = ent.get("pending")
        if pend is not None and pend() is not None:
            # an earlier forward of this recorded step has not been differentiated yet (and its autograd node is still
            # alive): its activations live in the recorded buffers, so this call must not touch them
            return self._forward(pixel_values, time, labels, pixel_mask, train)
        if ent["state"] == "ready":
            for dst, src in zip(ent["in"], ins):
                if dst is not None:
                    dst.copy_(src)
            self._replay(ent["fwd"], ent.get("fwd_c"))
            loss, pred, tape = ent["out"]
            # views of THIS recorded step's buffers, valid until its next replay: ScOT.forward hands the caller clones (the reference
            # returns fresh tensors; a rollout that keeps `hidden_states` across calls must not see them change)
            self.last_hidden, self.last_hidden_aliased = ent["hidden"], True
            if not train:          # nothing is kept for a backward: the recorded buffers are free again as soon as the outputs are copied
                return (None if loss is None else loss.clone()), pred.clone(), None
            tok = _Token()
            ent["pending"] = weakref.ref(tok)
            return loss.clone(), pred.clone(), dict(_ent=ent, _tok=tok)
        if ent["state"] != "warm":        # recorded forward whose backward never ran, or a tape that was switched off
            return self._forward(pixel_values, time, labels, pixel_mask, train)
        ent["in"] = tuple(None if t is None else t.clone() for t in ins)
        self._rec, self._rec_keep = [], []
        prev = ops.set_recorder(self._rec)
        try:
            loss, pred, tape = self._forward(*ent["in"], train)
        except BaseException:
            self._taped.pop(key, None)        # a partial recording is never replayed (and never compiled: the original error surfaces)
            raise
        finally:
            ops.set_recorder(prev)
            rec, keep = self._rec, self._rec_keep
            self._rec = self._rec_keep = None
        ent["fwd"], ent["keep"] = rec, keep
        ent["fwd_c"] = ops.compile_tape(rec) if self.tape_c else None
        ent["hidden"] = self.last_hidden
        if not train:
            ent["out"] = (loss, pred, None)
            ent["state"] = "ready"
            return (None if loss is None else loss.clone()), pred.clone(), None
        tape["_ent"] = ent
        ent["out"] = (loss, pred, tape)
        ent["state"] = "fwd"
        tok = _Token()
        ent["pending"] = weakref.ref(tok)
        return loss.clone(), pred.clone(), dict(_ent=ent, _tok=tok)

    def _capturing(self):
        return self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()

    def _stream_id(self):
        return torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0

    def _replay(self, cmds, compiled=None):
        timer = self.launch_timer
        if timer is not None:
            return self._replay_timed(cmds, timer)
        if compiled is not None:
            return ops.replay_tape(compiled)
        for fn, args in cmds:
            if args is None:
                fn()
            else:
                rc = fn(*args)
                if rc:
                    raise RuntimeError(f"step tape: {getattr(fn, '__name__', fn)} returned {rc}")

    def _replay_timed(self, cmds, timer):
        """Replay with a HIP-event pair around every C-ABI call, recorded on the stream the call launches on (its last
        argument) — bench.py's live per-kernel durations inside a real step (`engine.launch_timer = []` switches it on;
        entries are (entry point, arguments, start event, end event))."""
        streams = {}
        host_ops = ("scot_event_record", "scot_stream_wait_event", "scot_memset_async", "scot_memcpy_async")
        for fn, args in cmds:
            if args is None:
                fn()
                continue
            if getattr(fn, "__name__", "") in host_ops:       # stream / event plumbing and memsets: issued, not timed
                rc = fn(*args)
                if rc:
                    raise RuntimeError(f"step tape: {fn.__name__} returned {rc}")
                continue
            h = args[-1] or 0
            st = streams.get(h)
            if st is None:
                st = streams[h] = torch.cuda.ExternalStream(h) if h else torch.cuda.default_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            rc = fn(*args)
            e1.record(st)
            if rc:
                raise RuntimeError(f"step tape: {getattr(fn, '__name__', fn)} returned {rc}")
            timer.append((getattr(fn, "__name__", str(fn)), args, e0, e1))

    def _backward_step(self, tape, dloss=None, dpred=None):
        """Accumulates every parameter gradient into the gradient arena (+=).  dloss: [1] cuda tensor or None (=1)."""
        ent = tape.get("_ent")
        if ent is None:
            return self._backward(tape, dloss, dpred)
        ent["pending"] = None
        tape = ent["out"][2]     # the recorded step's activation records (the caller holds a per-call handle)
        if ent["state"] == "off":
            return self._backward(tape, dloss, dpred)
        if dpred is not None or self._capturing():   # not the recorded pattern: plain path, tape off
            ent["state"] = "off"
            return self._backward(tape, dloss, dpred)
        # one recorded backward per way the weight gradients meet the arena (first writers store after a lazy zero_grad, add otherwise:
        # the mode is an argument of the recorded launches)
        variant = bool(self.lazy_grads and self._small_chunks is not None)
        have = ent.setdefault("bwd", {}).get(variant)
        if ent["state"] == "ready" and have is not None:
            if dloss is None:
                ent["dloss"].fill_(1.0)
            else:
                ent["dloss"].copy_(dloss.reshape(1))
            self._lazy_now = variant
            self._replay(*have)
            return
        if ent["state"] not in ("fwd", "ready"):
            return self._backward(tape, dloss, dpred)
        if "dloss" not in ent:
            ent["dloss"] = torch.ones(1, device=self.device) if dloss is None else dloss.reshape(1).to(torch.float32).clone()
        elif dloss is None:
            ent["dloss"].fill_(1.0)
        else:
            ent["dloss"].copy_(dloss.reshape(1))
        self._rec, self._rec_keep = [], ent["keep"]
        prev = ops.set_recorder(self._rec)
        try:
            self._backward(tape, ent["dloss"], None)
        except BaseException:
            # never replayed: the signature is recorded afresh by its next steps (a failed forward recording already did that)
            self._taped = {k: v for k, v in self._taped.items() if v is not ent}
            ent["state"] = "off"
            import warnings
            warnings.warn("scOT engine: recording the backward of a step failed; the step tape of this input signature was dropped")
            raise
        finally:
            ops.set_recorder(prev)
            rec = self._rec
            self._rec = self._rec_keep = None
        ent["bwd"][variant] = (rec, ops.compile_tape(rec) if self.tape_c else None)
        ent["state"] = "ready"

    def reset_tapes(self):
        """Forget recorded steps (call after changing anything a tape bakes in: hooks, environment knobs)."""
        self._taped.clear()
        self._pool.clear()        # (recorded steps keep the rows they name alive themselves; inference with many shapes would pin one set each)

    def _forward(self, pixel_values, time=None, labels=None, pixel_mask=None, train=True, bool_masked_pos=None):
        cfg, cm = self.cfg, self.compute
        B, Cin, H, W = pixel_values.shape
        if Cin != cfg.num_channels:
            raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in the configuration.")
        if self.cond and time is None:
            raise ValueError("use_conditioning=True needs `time`")
        self.mark("fwd embed")
        p = cfg.patch_size
        gh, gw = self.grid
        C0 = cfg.embed_dim
        L0 = gh * gw
        tape = dict(B=B, time=time, enc=[], dec=[], res=[]) if train else None
        if train:
            self._init_grad_scale(B * cfg.num_out_channels * H * W)
        self.attn_sink = []
        ev_cpb = None

        def cpb_all():
            ops.cpb_fwd_batched(self.arena.data, self.cpb_desc, self.cpb_nlayers, self.cpb_max_ws, self.cpb_coords, self.cpb_tables,
                                self.cpb_z)
        if self.use_side and not self.stage_timing:
            _, ev_cpb = self.fork_task(cpb_all)     # the bias tables are first used by the first attention kernel: side stream
        else:
            cpb_all()
        # embeddings (model.py:295-366)
        cols = self.new(B * L0, Cin * p * p, dtype=self.tadt)
        ops.patchify(pixel_values, cols, B, Cin, H, W, p)
        e = self.new(B * L0, C0)
        wemb = self.TW("embeddings.patch_embeddings.projection.weight").view(C0, Cin * p * p)
        ops.linear_fwd(self.tcm, cols, wemb, e, bias=self.P("embeddings.patch_embeddings.projection.bias"))
        x, x16, est = self.norm_fwd("embeddings.norm", e, None, L0, C0, 1e-5, time, need_stats=train, copy=True)
        tokmask = None
        if bool_masked_pos is not None:      # model.py:353-359: masked positions take the learned mask token
            if "embeddings.mask_token" not in self.arena.offsets:
                raise ValueError("bool_masked_pos needs a model built with use_mask_token=True")
            tokmask = self.new(B * L0, dtype=torch.uint8)
            self.tdo(lambda: tokmask.copy_(bool_masked_pos.reshape(B * L0)))
            ops.mask_tokens(x, tokmask, self.P("embeddings.mask_token").view(-1), B * L0, C0)
            x16 = self.to_adt(x)
        if cfg.use_absolute_embeddings:
            ops.add(x, self.P("embeddings.position_embeddings").view(-1), x, period=L0 * C0)
            x16 = self.to_adt(x)
        if train:
            tape["emb"] = dict(cols=cols, e=e, stats=est, tokmask=tokmask)
        hidden_enc = [x]

        # encoder (model.py:816-861)
        skips: List[torch.Tensor] = []
        skip_ev = []
        side_skips = self.skip_side and self.use_side and not self.stage_timing

        def skip_blocks(i, st, s_in):
            """ConvNeXt blocks on skip i (model.py:1388-1393) → (processed skip, per-block records)"""
            nblk = int(cfg.skip_connections[i]) if i < len(cfg.skip_connections) else 0
            rr = []
            for j in range(nblk):
                s_in, r = self.convnext_fwd(f"residual_blocks.{i}.{j}", s_in, B, st.res[0], st.res[1], st.dim, time, train)
                rr.append(r)
            return s_in, rr
        res_recs = []
        self.wait_task(ev_cpb)
        for si, st in enumerate(self.enc):
            self.mark(f"fwd enc{si}")
            stage_in = x
            x, x16, recs = self.blocks_fwd(st.blocks, x, x16, B, time, train)
            skips.append(x)
            hidden_enc.append(x)
            if side_skips:      # this skip's blocks start now, beside the rest of the encoder
                nblk = int(cfg.skip_connections[si]) if si < len(cfg.skip_connections) else 0
                if nblk:
                    (skips[si], rr), ev = self.fork_task(lambda si=si, st=st, s_in=x: skip_blocks(si, st, s_in))
                else:
                    rr, ev = [], None
                res_recs.append(rr)
                skip_ev.append(ev)
            mrec = None
            if st.resample:
                x, x16, mrec = self.merge_fwd(st, x, stage_in, B, time, train)
            if train:
                tape["enc"].append((recs, mrec))

        self.mark("fwd convnext")
        if not side_skips:
            for i, st in enumerate(self.enc):
                skips[i], rr = skip_blocks(i, st, skips[i])
                res_recs.append(rr)
                skip_ev.append(None)
        if train:
            tape["res"] = res_recs

        # decoder (model.py:916-961, 1145-1240)
        self.wait_task(skip_ev[-1])
        x = skips[-1]
        x16 = self.to_adt(x)
        hidden_dec = [x]
        sk = skips[:-1]
        for k, st in enumerate(self.dec):
            self.mark(f"fwd dec{k}")
            if k != 0:
                self.wait_task(skip_ev[len(sk) - k])
                y = self.new(x.shape[0], x.shape[1])
                ops.add(x, sk[len(sk) - k], y)
                x = y
                x16 = self.to_adt(x)
            x, x16, recs = self.blocks_fwd(st.blocks, x, x16, B, time, train)
            hidden_dec.append(x)
            urec = None
            if st.resample:
                x, x16, urec = self.unmerge_fwd(st, x, x16, B, time, train)
            if train:
                tape["dec"].append((recs, urec))

        # recovery head (model.py:639-647)
        self.mark("fwd head")
        Cout = cfg.num_out_channels
        rc = self.new(B * L0, Cout * p * p)
        xr = x if self.tadt == torch.float32 else x16
        wrec = self.TW("patch_recovery.projection.weight").view(C0, Cout * p * p)
        ops.gemm(ops.NN, self.tcm, B * L0, Cout * p * p, C0, xr, C0, wrec, Cout * p * p, rc, Cout * p * p)
        img = self.new(B, Cout, H, W)
        ops.unpatchify(rc, self.P("patch_recovery.projection.bias"), img, B, Cout, H, W, gh, gw, p)
        pred = self.new(B, Cout, H, W)
        ops.conv5(img, self.P("patch_recovery.mixup.weight"), pred, B, Cout, H, W)

        # learn_residual, pixel_mask overwrite, loss (model.py:1411-1484)
        loss = None
        meta = self._loss_setup(Cout, B, H * W) if labels is not None else None
        pv_res = pixel_values if (cfg.learn_residual and self.cond) else None
        mask_u8, mask_full = None, False
        if pixel_mask is not None:
            if labels is None:
                raise ValueError("pixel_mask needs labels")
            mask_full = pixel_mask.dim() != 2
            mask_u8 = self.new(*((B, Cout, H, W) if mask_full else pixel_mask.shape), dtype=torch.uint8)
            self.tdo(lambda: mask_u8.copy_(pixel_mask.expand(B, Cout, H, W) if mask_full else pixel_mask))
        sums = None
        if labels is not None or pv_res is not None:
            sums = self.zeros(2 * meta["G"]) if meta else None
            ops.head_finalize(pred, pv_res, Cin, labels, mask_u8, mask_full, meta["goc"] if meta else None, sums, B, Cout, H * W,
                              cfg.p)
        if labels is not None:
            loss = self.new(1)
            ops.loss_finish(sums, meta["counts"], meta["G"], meta["normalized"], loss)
        if train:
            tape["head"] = dict(x=xr, img=img, pred=pred, labels=labels, mask=mask_u8, mask_full=mask_full, sums=sums, meta=meta,
                                shape=(B, Cout, H, W))
            tape["hidden"] = (hidden_dec, hidden_enc)
        self.last_hidden = (hidden_dec, hidden_enc)
        self.last_hidden_aliased = self._rec is not None      # rows of a step being recorded: its replays overwrite them
        return loss, pred, tape

    def _loss_setup(self, Cout, B, HW):
        cfg = self.cfg
        if cfg.p not in (1, 2):
            raise ValueError("p must be 1 or 2")
        groups = cfg.channel_slice_list_normalized_loss
        key = (Cout, B, HW, tuple(groups) if groups else None)
        if self._loss_meta and self._loss_meta[0] == key:
            if self._rec is not None:
                self._rec_keep.append(self._loss_meta[1])     # a recorded step names these tensors' addresses: it keeps them alive
            return self._loss_meta[1]
        goc = torch.full((Cout,), -1, dtype=torch.int32)
        if groups:
            G = len(groups) - 1
            counts = torch.zeros(G)
            for g in range(G):
                goc[groups[g]:groups[g + 1]] = g
                counts[g] = B * (groups[g + 1] - groups[g]) * HW
        else:
            G = 1
            goc[:] = 0
            counts = torch.tensor([float(B * Cout * HW)])
        meta = dict(G=G, goc=goc.to(self.device), counts=counts.to(self.device), normalized=bool(groups))
        self._loss_meta = (key, meta)
        if self._rec is not None:
            self._rec_keep.append(meta)       # (the one-entry cache above is replaced when another batch size comes along)
        return meta

    def _backward(self, tape, dloss=None, dpred=None):
        cfg, cm, adt = self.cfg, self.compute, self.adt
        B, time = tape["B"], tape["time"]
        hd = tape["head"]
        self._lazy_now = bool(self.lazy_grads and self._small_chunks is not None)
        def fill_done():        # ScOT.zero_grad(overlap=True): the arena's fill runs on the side stream beside the forward
            ev, self.grad_fill_event = self.grad_fill_event, None
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
        self.tdo_dynamic(fill_done)
        self.h_zero(self.cpb_dtables)
        if self.cpb_dls is not None:
            self.h_zero(self.cpb_dls)
        self.mark("bwd head")
        _, Cout, H, W = hd["shape"]
        p = cfg.patch_size
        gh, gw = self.grid
        C0, L0 = cfg.embed_dim, gh * gw
        # gradient scale of the fp16 build (1.0 otherwise): gradients already in the arena are brought to the same scale first
        scaled = self.scale_grads
        if scaled:
            S_dev, Sinv_dev = self.scale_state[0:1], self.scale_state[1:2]

            def prescale():
                if not self.grads_are_zero:
                    self.scale_grad_range(S_dev)
            self.tdo_dynamic(prescale)
            if dpred is not None:
                dpred = self.clone(dpred.contiguous())
                ops.scale_inplace_dev(dpred.view(-1), S_dev)
            if hd["labels"] is not None:
                dl_in = dloss
                dloss = self.new(1)
                if dl_in is None:
                    self.h_copy(dloss, S_dev)
                else:
                    self.tdo(lambda: torch.mul(dl_in.reshape(1), S_dev, out=dloss))
        # loss → d pred
        if hd["labels"] is not None:
            g_pred = self.new(B, Cout, H, W)
            meta = hd["meta"]
            ops.loss_bwd(hd["pred"], hd["labels"], hd["mask"], hd["mask_full"], meta["goc"], hd["sums"], meta["counts"], meta["G"],
                         meta["normalized"], dloss, g_pred, B, Cout, H * W, cfg.p)
            if dpred is not None:
                ops.add(g_pred, dpred.contiguous(), g_pred)
        elif dpred is not None:
            g_pred = self.clone(dpred.contiguous())
        else:
            raise RuntimeError("nothing to differentiate: no labels and no gradient for the prediction")
        # recovery head
        self.off_critical_path(lambda: ops.conv5_wgrad(g_pred, hd["img"], self.G("patch_recovery.mixup.weight"), B, Cout, H, W),
                               g_pred, hd["img"])
        d_img = self.new(B, Cout, H, W)
        ops.conv5(g_pred, self.P("patch_recovery.mixup.weight"), d_img, B, Cout, H, W, transpose=True)
        self.off_critical_path(lambda: ops.nchw_channel_sum(d_img, self.G("patch_recovery.projection.bias"), B, Cout, H * W), d_img)
        d_rc = self.new(B * L0, Cout * p * p, dtype=self.tadt)
        ops.patchify(d_img, d_rc, B, Cout, H, W, p)
        wrec = self.TW("patch_recovery.projection.weight").view(C0, Cout * p * p)
        self.off_critical_path(lambda: ops.gemm(ops.TN, self.tcm, C0, Cout * p * p, B * L0, hd["x"], C0, d_rc, Cout * p * p,
                                                self.G("patch_recovery.projection.weight").view(C0, Cout * p * p), Cout * p * p,
                                                accumulate=True), hd["x"], d_rc)
        g = self.new(B * L0, C0)
        ops.gemm(ops.NT, self.tcm, B * L0, C0, Cout * p * p, d_rc, Cout * p * p, wrec, Cout * p * p, g, C0)
        if self.on_grads_final is not None:
            from .dp import group_ranges

            def announce(prefix, _cb=self.on_grads_final):
                # the callback runs with the SIDE stream current: what it enqueues (dp.py: an event the comm stream waits for) is
                # ordered behind the range's weight gradients and un-scale without the main chain ever waiting for them
                with torch.cuda.stream(self.side):
                    _cb(prefix)

            def done(prefix, _cb=self.on_grads_final):
                if not self.use_side:
                    self.flush_side()  # (no side stream: the range's queued weight gradients run here, in line)
                    if scaled:         # back at scale 1 before the range goes on the wire
                        self.scale_grad_range(Sinv_dev, prefix, count=True)
                    # (the callback launches through ops — pack / collective / unpack: with the recorder left on, those launches would
                    # be logged IN ADDITION to the callback itself and a replayed step would run them twice)
                    self.tdo_dynamic(lambda: _cb(prefix))
                    return
                # side stream (forked behind everything the main chain has enqueued so far, i.e. behind the range's last
                # main-stream gradient kernel): the range's queued weight gradients, its un-scale, then the announcement
                self.flush_side()
                if scaled:
                    self.off_critical_path(lambda: self.scale_grad_range(Sinv_dev, prefix, count=True))
                self.off_critical_path(lambda: self.tdo_dynamic(lambda: announce(prefix)))
                self.flush_side()
        elif scaled and self.use_side:
            from .dp import group_ranges

            def done(prefix):
                # fp16 build: bring each range back from the gradient scale as soon as the backward has finished writing it, on
                # the side stream behind the range's weight gradients (one 0.24 ms pass at the very end of the step before)
                self.flush_side()          # the range's queued weight gradients go first
                self.off_critical_path(lambda: self.scale_grad_range(Sinv_dev, prefix, count=True))
                self.flush_side()
        else:
            def done(prefix):
                return None
        _range_done = done

        def done(prefix):       # a range is only final once the queued partial sums of its norms have been added in
            self.finish_partials()
            _range_done(prefix)
        done("patch_recovery.")
        from .dp import stage_groups, stage_split

        def stage_bwd(prefix, st, recs, g):
            """the blocks of one stage; a stage announced in halves (dp.stage_split) hands over blocks[s:] as soon as block s is done"""
            s_ = stage_split(cfg, prefix)
            keys = stage_groups(cfg, prefix)
            if not s_:
                g = self.blocks_bwd(recs, g, B, time)
                self.cpb_backward_range(st.blocks)
                done(keys[0])
                return g

            def upper():
                self.cpb_backward_range(st.blocks[s_:])
                done(keys[0])
            g = self.blocks_bwd(recs, g, B, time, split=s_, on_upper_half=upper)
            self.cpb_backward_range(st.blocks[:s_])
            done(keys[1])
            return g

        # decoder, shallow → deep
        nl = len(self.dec)
        g_skips: List[Optional[torch.Tensor]] = [None] * nl  # gradient wrt the (ConvNeXt-processed) skips
        skip_ev = [None] * nl
        side_skips = self.skip_side and self.use_side and not self.stage_timing

        def skip_bwd(i, gi):
            """backward of the ConvNeXt blocks on skip i: gradient wrt the encoder stage's output (in place on gi)"""
            st_ = self.enc[i]
            for j in reversed(range(len(tape["res"][i]))):
                gi = self.convnext_bwd(f"residual_blocks.{i}.{j}", tape["res"][i][j], gi, B, st_.res[0], st_.res[1], st_.dim, time)
            return gi
        for k in reversed(range(nl)):
            self.mark(f"bwd dec{k}")
            st = self.dec[k]
            recs, urec = tape["dec"][k]
            if st.resample:
                g = self.unmerge_bwd(st, urec, g, B, time)
            g = stage_bwd(f"decoder.layers.{k}.", st, recs, g)
            if k != 0:
                i = nl - 1 - k
                g_skips[i] = g   # x = x_prev + skip: both get g (g keeps flowing to x_prev unchanged)
                g = self.clone(g)
                if side_skips and tape["res"][i]:
                    # this skip's gradient is only needed when the backward reaches encoder stage i: its ConvNeXt blocks go to
                    # the side stream now, beside the deeper decoder / encoder stages
                    g_skips[i], skip_ev[i] = self.fork_task(lambda i=i, gi=g_skips[i]: skip_bwd(i, gi))
        g_skips[nl - 1] = g  # decoder input = skips[-1]

        # ConvNeXt blocks (in line: the deepest skip, and every skip when the side stream is off)
        self.mark("bwd convnext")
        for i in reversed(range(nl)):
            if skip_ev[i] is None:
                g_skips[i] = skip_bwd(i, g_skips[i])

        if not side_skips:
            done("residual_blocks.")
        # encoder, deep → shallow
        g = None
        for s in reversed(range(nl)):
            self.mark(f"bwd enc{s}")
            st = self.enc[s]
            recs, mrec = tape["enc"][s]
            self.wait_task(skip_ev[s])
            if st.resample:
                d_sum = self.merge_bwd(st, mrec, g, B, time)
                g = g_skips[s]
                ops.add(g, d_sum, g)
            else:
                g = g_skips[s]
                d_sum = None
            g = stage_bwd(f"encoder.layers.{s}.", st, recs, g)
            if d_sum is not None:
                ops.add(g, d_sum, g)
        if side_skips:
            done("residual_blocks.")     # (their side-stream tasks were joined by the encoder stages that consumed them)

        # embeddings
        self.mark("bwd embed")
        emb = tape["emb"]
        Cin = cfg.num_channels
        if cfg.use_absolute_embeddings:
            ops.batch_sum(g, self.G("embeddings.position_embeddings").view(-1), B, L0 * C0)
        if emb.get("tokmask") is not None:
            ops.mask_tokens_bwd(g, emb["tokmask"], self.G("embeddings.mask_token").view(-1), B * L0, C0)
        d_e = self.norm_bwd("embeddings.norm", g, emb["e"], emb["stats"], L0, C0, time, self.tadt)
        self.wgrad(self.tcm, d_e, emb["cols"], self.G("embeddings.patch_embeddings.projection.weight").view(C0, Cin * p * p),
                   dbias=self.G("embeddings.patch_embeddings.projection.bias"))
        if scaled and self.on_grads_final is None and not self.use_side:
            self.join_side()
            self.scale_grad_range(Sinv_dev, count=True)
        self.mark("end")
        done("embeddings.")
        self.join_side()
