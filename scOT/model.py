"""scOT.model — the reference's model API on the MI355X-native engine.

Public surface mirrored from the reference (reference scOT/model.py):
  ScOTConfig, ScOTOutput, LayerNorm, ConditionalLayerNorm, ScOT(config, use_mask_token=False),
  ScOT.forward(pixel_values, time, bool_masked_pos, head_mask, pixel_mask, labels, output_attentions,
               output_hidden_states, return_dict) -> ScOTOutput | tuple,
  ScOT.from_pretrained / save_pretrained (HF checkpoint directory: config.json + model.safetensors | pytorch_model.bin),
  state_dict keys / shapes identical to the reference.

The module tree below only OWNS parameters (as views into a flat arena, poseidon_amd/arena.py) under the reference's
names; it contains no arithmetic.  `forward` hands pointers to poseidon_amd.engine.ScOTEngine, whose every op is a
hand-written HIP kernel.  There is no CPU path: calling forward on CPU tensors raises.
"""
from __future__ import annotations

import json
import math
import os
from collections import OrderedDict
from dataclasses import dataclass, fields
from typing import Optional, Tuple

import torch
from torch import nn

from poseidon_amd.arena import Arena
from poseidon_amd.config import MODEL_MAP, ScOTConfig, preset  # noqa: F401  (re-exported)
from poseidon_amd.engine import ScOTEngine
from poseidon_amd.geometry import param_shapes, stage_plan
from poseidon_amd.lib import ScotLibraryError


@dataclass
class ScOTOutput:
    """reference model.py:57-63; indexable like HF ModelOutput (`out["loss"]`, `out[0]`, `.output`)."""
    loss: Optional[torch.Tensor] = None
    output: Optional[torch.Tensor] = None
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None
    attentions: Optional[Tuple[torch.Tensor, ...]] = None
    reshaped_hidden_states: Optional[Tuple[torch.Tensor, ...]] = None

    def to_tuple(self):
        return tuple(getattr(self, f.name) for f in fields(self) if getattr(self, f.name) is not None)

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        return self.to_tuple()[k]

    def keys(self):
        return [f.name for f in fields(self) if getattr(self, f.name) is not None]

    def __contains__(self, k):
        return k in self.keys()

    def items(self):
        return [(k, getattr(self, k)) for k in self.keys()]

    def values(self):
        return [getattr(self, k) for k in self.keys()]


def _require_hip(t: torch.Tensor) -> None:
    """The hot path is HIP-only: no CPU fallback exists in the product.  (tests/hipemu, which executes the kernel sources on the
    host, replaces this check together with `ops.ptr` for the duration of a test.)"""
    if not t.is_cuda:
        raise ScotLibraryError("ScOT.forward needs CUDA(HIP) tensors: the hot path is HIP-only (no CPU fallback)")


# ---------------------------------------------------------------------------------------------- parameter containers
class LayerNorm(nn.LayerNorm):
    """Time-independent variant (reference model.py:135-140): parameters `weight`, `bias` of shape (C,)."""

    def forward(self, x, time=None):  # pragma: no cover - containers hold parameters only
        raise ScotLibraryError("normalisation runs inside the fused HIP kernels (scot_cln_fwd); call ScOT.forward")


class ConditionalLayerNorm(nn.Module):
    """Time-conditioned layer norm (reference model.py:143-160): gamma = weight(t), beta = bias(t), two nn.Linear(1, C)."""

    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Linear(1, dim)
        self.bias = nn.Linear(1, dim)

    def forward(self, x, time):  # pragma: no cover
        raise ScotLibraryError("normalisation runs inside the fused HIP kernels (scot_cln_fwd); call ScOT.forward")


def _norm(cfg, dim, eps=1e-5):
    return ConditionalLayerNorm(dim, eps=eps) if cfg.use_conditioning else LayerNorm(dim, eps=eps)


class _Holder(nn.Module):
    """Plain namespace module (children registered by attribute assignment)."""


def _attention(cfg, dim, heads):
    att = _Holder()
    att.self = _Holder()
    att.self.logit_scale = nn.Parameter(torch.log(10 * torch.ones((heads, 1, 1))))  # HF:374
    att.self.continuous_position_bias_mlp = nn.Sequential(nn.Linear(2, 512, bias=True), nn.ReLU(inplace=True),
                                                          nn.Linear(512, heads, bias=False))
    att.self.query = nn.Linear(dim, dim, bias=cfg.qkv_bias)
    att.self.key = nn.Linear(dim, dim, bias=False)
    att.self.value = nn.Linear(dim, dim, bias=cfg.qkv_bias)
    att.output = _Holder()
    att.output.dense = nn.Linear(dim, dim)
    return att


def _block(cfg, dim, heads):
    b = _Holder()
    b.attention = _attention(cfg, dim, heads)
    b.layernorm_before = _norm(cfg, dim, cfg.layer_norm_eps)
    b.intermediate = _Holder()
    b.intermediate.dense = nn.Linear(dim, int(cfg.mlp_ratio * dim))
    b.output = _Holder()
    b.output.dense = nn.Linear(int(cfg.mlp_ratio * dim), dim)
    b.layernorm_after = _norm(cfg, dim, cfg.layer_norm_eps)
    return b


def _convnext(cfg, dim):
    c = _Holder()
    c.weight = nn.Parameter(1e-6 * torch.ones(dim))  # layer scale (reference model.py:191-195)
    c.dwconv = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
    c.norm = _norm(cfg, dim, cfg.layer_norm_eps)
    c.pwconv1 = nn.Linear(dim, 4 * dim)
    c.pwconv2 = nn.Linear(4 * dim, dim)
    return c


class _ScOTFunction(torch.autograd.Function):
    """Connects the engine's explicit forward/backward to torch.autograd.  Parameter gradients are accumulated by the
    kernels directly into the gradient arena whose slices are the parameters' `.grad` (no per-tensor AccumulateGrad)."""

    @staticmethod
    def forward(ctx, anchor, model, pixel_values, time, labels, pixel_mask, bool_masked_pos=None):
        # activations are kept because a gradient was asked for; stochastic depth follows module.training (HF:565-586)
        loss, pred, tape = model._engine.forward(pixel_values, time, labels, pixel_mask, train=True, stochastic=model.training,
                                                 bool_masked_pos=bool_masked_pos)
        ctx.model, ctx.tape = model, tape
        ctx.has_loss = loss is not None
        ctx.set_materialize_grads(False)  # unused outputs arrive as None instead of zero tensors
        if loss is None:
            loss = pred.new_zeros(1)
        return loss.view(()), pred

    @staticmethod
    def backward(ctx, dloss, dpred):
        model, tape = ctx.model, ctx.tape
        ctx.tape = None
        if tape is None:
            raise RuntimeError("ScOT backward called twice (activations are freed after the first backward)")
        model._prepare_grads()
        dl = None
        if ctx.has_loss:
            dl = (dloss.reshape(1).to(torch.float32).contiguous() if dloss is not None
                  else torch.zeros(1, device=model._arena.data.device))
        model._engine.backward(tape, dl, dpred)
        model._after_backward()
        return None, None, None, None, None, None, None


class ScOT(nn.Module):
    config_class = ScOTConfig
    base_model_prefix = "swinv2"
    main_input_name = "pixel_values"

    def __init__(self, config: ScOTConfig, use_mask_token: bool = False, compute: Optional[str] = None, engine_options: Optional[dict] = None):
        super().__init__()
        self.engine_options = dict(engine_options or {})      # overrides of poseidon_amd.engine.ENGINE_OPTIONS for this model's engine
        if config.residual_model not in ("convnext", "resnet"):
            raise ValueError("residual_model must be 'convnext' or 'resnet'")
        if config.residual_model != "convnext":
            raise NotImplementedError("residual_model='resnet' is unused by every preset and out of scope (SURVEY.md §8a row 18)")
        if config.image_size % config.patch_size:
            raise ValueError("image_size must be a multiple of patch_size")
        if config.hidden_dropout_prob or config.attention_probs_dropout_prob:
            raise NotImplementedError("hidden_dropout_prob / attention_probs_dropout_prob != 0 are not implemented (every preset "
                                      "and the training recipe use 0.0, reference train.py:247-272); refusing to ignore them silently")
        self.config = config
        # default: IEEE binary16 MFMA operands with fp32 accumulation / statistics / residual stream — the fastest mode whose
        # forward stays within the north star's 1e-3 of the fp32 reference (7e-4 on trained-like parameters);
        # "fp32" (exact fp32 MFMA, 1e-6), "bf16x3" (1e-5) and "bf16" (6e-3: NOT within the bound) are opt-in
        self.compute = compute or os.environ.get("SCOT_COMPUTE", "fp16")
        self.num_layers_encoder = self.num_layers_decoder = len(config.depths)
        self.num_features = int(config.embed_dim * 2 ** (len(config.depths) - 1))
        cfg = config
        c0, p = cfg.embed_dim, cfg.patch_size
        grid, enc, dec = stage_plan(cfg)

        self.embeddings = _Holder()
        if use_mask_token:      # reference model.py:323-327 (own parameters of `embeddings` register before its children)
            self.embeddings.mask_token = nn.Parameter(torch.zeros(1, 1, c0))
        if cfg.use_absolute_embeddings:
            self.embeddings.position_embeddings = nn.Parameter(torch.zeros(1, grid[0] * grid[1], c0))
        self.embeddings.patch_embeddings = _Holder()
        self.embeddings.patch_embeddings.projection = nn.Conv2d(cfg.num_channels, c0, kernel_size=p, stride=p)
        self.embeddings.norm = _norm(cfg, c0)

        self.encoder = _Holder()
        self.encoder.layers = nn.ModuleList()
        for st in enc:
            s = _Holder()
            s.blocks = nn.ModuleList([_block(cfg, st.dim, st.heads) for _ in st.blocks])
            if st.resample:
                s.downsample = _Holder()
                s.downsample.reduction = nn.Linear(4 * st.dim, 2 * st.dim, bias=False)
                s.downsample.norm = _norm(cfg, 2 * st.dim)
            self.encoder.layers.append(s)
        self.decoder = _Holder()
        self.decoder.layers = nn.ModuleList()
        for st in dec:
            s = _Holder()
            s.blocks = nn.ModuleList([_block(cfg, st.dim, st.heads) for _ in st.blocks])
            if st.resample:
                s.upsample = _Holder()
                s.upsample.upsample = nn.Linear(st.dim, 2 * st.dim, bias=False)
                s.upsample.mixup = nn.Linear(st.dim // 2, st.dim // 2, bias=False)
                s.upsample.norm = _norm(cfg, st.dim // 2)
            self.decoder.layers.append(s)
        self.patch_recovery = _Holder()
        self.patch_recovery.projection = nn.ConvTranspose2d(c0, cfg.num_out_channels, kernel_size=p, stride=p)
        self.patch_recovery.mixup = nn.Conv2d(cfg.num_out_channels, cfg.num_out_channels, kernel_size=5, stride=1, padding=2,
                                              bias=False)
        self.residual_blocks = nn.ModuleList([
            nn.ModuleList([_convnext(cfg, cfg.embed_dim * 2 ** i) for _ in range(int(depth))]) if int(depth) > 0
            else nn.ModuleList([nn.Identity()]) for i, depth in enumerate(cfg.skip_connections)])

        self._init_weights()
        self.use_mask_token = bool(use_mask_token)
        self._shapes = param_shapes(cfg, use_mask_token=self.use_mask_token)
        got = OrderedDict((k, tuple(v.shape)) for k, v in self.named_parameters())
        if list(got.items()) != list(self._shapes.items()):  # schema self-check
            raise AssertionError("parameter schema drifted from poseidon_amd.geometry.param_shapes")
        self._arena: Optional[Arena] = None
        self._explicit_version = 0
        self._engine: Optional[ScOTEngine] = None
        self._anchor = None
        self._grad_hooks = []

    # ------------------------------------------------------------------------------------------ init / io
    def _init_weights(self):
        """HF `_init_weights` semantics of the pinned stack: Linear/Conv2d ~ N(0, initializer_range),
        biases 0, LayerNorm 1/0; logit_scale, layer-scale and the ConvTranspose2d keep their constructor values."""
        std = self.config.initializer_range
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                m.weight.data.normal_(mean=0.0, std=std)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.LayerNorm):
                m.bias.data.zero_()
                m.weight.data.fill_(1.0)

    def get_input_embeddings(self):
        return self.embeddings.patch_embeddings

    def _prune_heads(self, heads_to_prune):
        """reference model.py:1287-1291 (HF head pruning).  The q/k/v weights of a block live back to back in the parameter arena
        and the attention kernels are compiled for head_dim = embed_dim / 3: removing heads changes both.  No Poseidon recipe
        prunes heads; refuse instead of silently computing with the unpruned model."""
        raise NotImplementedError("head pruning is not supported by the fused attention kernels (reference model.py:1287-1291)")

    def prune_heads(self, heads_to_prune):
        self._prune_heads(heads_to_prune)

    def num_parameters(self):
        return sum(p.numel() for p in self.parameters())

    def push_to_hub(self, repo_id: str, **kwargs):
        """HF `PreTrainedModel.push_to_hub` (reference train.py:413): the checkpoint directory `save_pretrained` writes, uploaded with
        huggingface_hub (needs network and credentials; nothing of it runs on the GPU path)."""
        import tempfile
        from huggingface_hub import HfApi
        api = HfApi(token=kwargs.pop("token", None))
        api.create_repo(repo_id, exist_ok=True, private=kwargs.pop("private", None))
        with tempfile.TemporaryDirectory() as d:
            self.save_pretrained(d, safe_serialization=kwargs.pop("safe_serialization", True))
            return api.upload_folder(repo_id=repo_id, folder_path=d, commit_message=kwargs.pop("commit_message", "Upload model"))

    def save_pretrained(self, save_directory: str, safe_serialization: bool = True, **_):
        os.makedirs(save_directory, exist_ok=True)
        self.config.save_pretrained(save_directory)
        sd = {k: v.detach().to("cpu").contiguous().clone() for k, v in self.state_dict().items()}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, "model.safetensors"), metadata={"format": "pt"})
        else:
            torch.save(sd, os.path.join(save_directory, "pytorch_model.bin"))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, config: Optional[ScOTConfig] = None,
                        ignore_mismatched_sizes: bool = False, compute: Optional[str] = None, **kwargs):
        path = pretrained_model_name_or_path
        if not os.path.isdir(path):
            # a hub id (reference train.py:331-333: "camlab-ethz/Poseidon-B"): resolved through huggingface_hub — from its local cache
            # first, from the network when there is one (`local_files_only` / `cache_dir` / `revision` / `token` are passed on)
            try:
                from huggingface_hub import snapshot_download
                hub_kw = {k: kwargs[k] for k in ("cache_dir", "revision", "token", "local_files_only") if k in kwargs}
                try:
                    path = snapshot_download(path, allow_patterns=["*.json", "*.safetensors", "*.bin"], **{"local_files_only": True, **hub_kw})
                except Exception:
                    if hub_kw.get("local_files_only"):
                        raise
                    path = snapshot_download(path, allow_patterns=["*.json", "*.safetensors", "*.bin"], **hub_kw)
            except Exception as e:
                raise FileNotFoundError(f"{pretrained_model_name_or_path}: neither a checkpoint directory nor a hub repository that "
                                        f"huggingface_hub can resolve here ({type(e).__name__}: {e}); download `camlab-ethz/Poseidon-*` "
                                        "and pass the directory") from e
        if config is None:
            config = ScOTConfig.from_pretrained(path)
        model = cls(config, compute=compute)
        st, bn = os.path.join(path, "model.safetensors"), os.path.join(path, "pytorch_model.bin")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        elif os.path.exists(bn):
            sd = torch.load(bn, map_location="cpu", weights_only=True)
        else:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin in {path}")
        own = model.state_dict()
        mism, missing = [], [k for k in own if k not in sd]
        unexpected = [k for k in sd if k not in own]
        for k, v in sd.items():
            if k in own:
                if tuple(v.shape) != tuple(own[k].shape):
                    if not ignore_mismatched_sizes:
                        raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(own[k].shape)} "
                                           "(pass ignore_mismatched_sizes=True to re-initialise, reference train.py:331-333)")
                    mism.append(k)
                else:
                    own[k].copy_(v)
        model._load_report = dict(missing=missing, unexpected=unexpected, mismatched=mism)
        return model

    # ------------------------------------------------------------------------------------------ arena management
    def _ensure_arena(self, device):
        ar = self._arena
        if ar is not None and ar.data.device == device:
            # the parameters still are views of the arena (first and last one checked; `_params` is the list the arena was built from: walking
            # the module tree with named_parameters() on EVERY forward cost 4 ms of host time per step of Poseidon-B)
            ps, gv = self._params, self._pviews
            if ps[0].data_ptr() == gv[0] and ps[-1].data_ptr() == gv[1]:
                return
        params = list(self.named_parameters())
        ar = Arena(self._shapes, device)
        with torch.no_grad():
            for n, p in params:
                v = ar.view(n)
                v.copy_(p.data.to(device=device, dtype=torch.float32))
                p.data = v
                p.grad = None
        self._arena = ar
        self._engine = ScOTEngine(self.config, ar, self.compute, options=self.engine_options)
        self._engine.weights_version = self._weights_version
        self._anchor = torch.zeros(1, device=device, requires_grad=True)
        self._params = [p for _, p in params]
        self._pviews = (ar.view(params[0][0]).data_ptr(), ar.view(params[-1][0]).data_ptr())
        self._gviews = [ar.gview(n) for n, _ in params]

    def _weights_version(self):
        """Moves whenever the fp32 master weights may have changed: in-place torch ops on a parameter or on the flat arena (optimizers,
        load_state_dict, `p.add_()` under no_grad) bump torch's version counters; everything that writes behind torch's back — kernels
        through raw pointers (the fused AdamW), code that writes through `p.data`, and COLLECTIVES (`dist.broadcast` / `all_reduce`
        leave `tensor._version` unchanged: `GradAllReducer.broadcast_parameters` does it) — calls `mark_weights_dirty()`.  The engine
        re-casts its 16-bit weight copies only when this value differs from the one they were made from."""
        return (sum(p._version for p in self._params), self._arena.data._version, self._explicit_version)

    def mark_weights_dirty(self):
        """Tell the engine the parameters were modified behind torch's back (writes through `p.data`, foreign kernels)."""
        self._explicit_version += 1

    def flat_parameters(self) -> torch.Tensor:
        return self._arena.data

    def flat_grads(self) -> torch.Tensor:
        return self._arena.grad

    def _prepare_grads(self):
        """Attach arena slices as `.grad`.  If the grads were set to None (zero_grad(set_to_none=True)) the arena is
        cleared first, so accumulation semantics match autograd's."""
        ps = self._params
        first = next((i for i, p in enumerate(ps) if p.requires_grad), None)   # a frozen parameter's .grad is always None
        if first is None:
            return
        if ps[first].grad is None or ps[first].grad.data_ptr() != self._gviews[first].data_ptr():
            self._arena.grad.zero_()
            self._engine.grads_are_zero = True
            self._engine.lazy_grads = False
            for p, g in zip(ps, self._gviews):
                p.grad = g if p.requires_grad else None

    def _after_backward(self):
        for h in self._grad_hooks:
            h(self)

    def register_grad_ready_hook(self, fn):
        """fn(model) is called right after the engine finished writing the gradient arena (used by the DP wrapper)."""
        self._grad_hooks.append(fn)

    def zero_grad(self, set_to_none: bool = False, overlap: bool = False, lazy: Optional[bool] = None):
        """One fill of the gradient arena; `.grad` stay views of it.  overlap=True (the training loops of this library): the fill
        is queued on the engine's weight-gradient stream behind everything the current stream has been given so far, and the next
        backward's first gradient writer waits for it — the next FORWARD does not (it never touches gradients), so the fill
        runs beside it instead of in front of it.  Code that reads `.grad` on the current stream between this call and
        the next backward must use the default (ordered on the current stream).
        lazy (default: = overlap; 16-bit compute modes): the Linear weights of the ScOTLayers — 95 % of the bytes — are NOT filled; the
        next backward's weight-gradient kernels store into them instead of accumulating (engine.lazy_grads), so between this call and
        that backward those `.grad` still hold the previous step's values: the contract of the overlapped form already (nobody may
        read `.grad` in between), spelled out."""
        if self._arena is not None:
            eng = self._engine
            lazy = bool(overlap if lazy is None else lazy) and eng is not None and eng._small_chunks is not None
            fill = eng.fill_small_grads if lazy else self._arena.grad.zero_
            if overlap and eng is not None and eng.use_side and self._arena.grad.is_cuda:
                cur, side = torch.cuda.current_stream(), eng.side_stream()
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    fill()
                ev = torch.cuda.Event()
                ev.record(side)
                eng.grad_fill_event = ev
            else:
                fill()
            self._engine.grads_are_zero = True
            self._engine.lazy_grads = lazy
            if set_to_none:
                for p in self._params:
                    p.grad = None
        else:
            super().zero_grad(set_to_none=set_to_none)

    # ------------------------------------------------------------------------------------------ forward
    @staticmethod
    def _downsample(image, target_size):
        """Spectral down-sampling (reference model.py:1293-1300) on the native kernels (see spectral_resize)."""
        return spectral_resize(image, target_size)

    @staticmethod
    def _upsample(image, target_size):
        """Spectral up-sampling (reference model.py:1302-1316)."""
        return spectral_resize(image, target_size)

    def forward(self, pixel_values=None, time=None, bool_masked_pos=None, head_mask=None, pixel_mask=None, labels=None,
                output_attentions=None, output_hidden_states=None, return_dict=None):
        cfg = self.config
        return_dict = return_dict if return_dict is not None else cfg.use_return_dict
        if pixel_values is None:
            raise ValueError("pixel_values cannot be None")
        want_attn = bool(output_attentions or (output_attentions is None and getattr(cfg, "output_attentions", False)))
        if bool_masked_pos is not None and not self.use_mask_token:
            raise ValueError("bool_masked_pos needs ScOT(config, use_mask_token=True) (reference model.py:323-327, 353-359)")
        if head_mask is not None:
            raise NotImplementedError("head_mask is not implemented")
        _require_hip(pixel_values)
        dev = pixel_values.device
        self._ensure_arena(dev)
        pv = pixel_values.to(torch.float32).contiguous()
        B = pv.shape[0]
        t = None
        if cfg.use_conditioning:
            if time is None:
                raise ValueError("time is required when use_conditioning=True")
            t = torch.as_tensor(time, device=dev).reshape(-1).to(torch.float32).contiguous()
            if t.numel() == 1 and B > 1:
                t = t.expand(B).contiguous()
        lab = labels.to(device=dev, dtype=torch.float32).contiguous() if labels is not None else None
        in_size = pv.shape[2]
        resized = in_size != cfg.image_size
        if resized:
            pv = self._upsample(pv, cfg.image_size) if in_size < cfg.image_size else self._downsample(pv, cfg.image_size)
        want_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self._params)
        bmp = None
        if bool_masked_pos is not None:
            if resized:
                raise ValueError("bool_masked_pos indexes the patch grid of config.image_size: no spectral resize with it")
            bmp = bool_masked_pos.to(device=dev).reshape(B, -1).contiguous()
        self._engine.collect_attn = want_attn     # (read by the engine during this call only; reset below)
        if resized and (lab is not None or pixel_mask is not None):
            # reference order (model.py:1416-1484): resize the prediction back first, then mask + loss at input resolution
            loss, pred = self._forward_resized(pv, t, lab, pixel_mask, in_size, want_grad)
        elif want_grad:
            loss, pred = _ScOTFunction.apply(self._anchor, self, pv, t, lab, pixel_mask, bmp)
            if lab is None:
                loss = None
        else:
            loss, pred, _ = self._engine.forward(pv, t, lab, pixel_mask, train=False, stochastic=self.training, bool_masked_pos=bmp)
            if loss is not None:
                loss = loss.view(())
            if resized:
                pred = self._upsample(pred, in_size) if in_size > cfg.image_size else self._downsample(pred, in_size)
        self._engine.collect_attn = False
        hs = rhs = None
        want_hs = bool(output_hidden_states or (output_hidden_states is None and cfg.output_hidden_states))
        if want_hs or not return_dict:
            hd, he = self._engine.last_hidden
            if self._engine.last_hidden_aliased:      # rows of a recorded step (overwritten by its next replay): the caller gets its own
                hd, he = [h.clone() for h in hd], [h.clone() for h in he]
            hs, rhs, enc_hs, enc_rhs, dec_hs, dec_rhs = self._hidden_tuples(hd, he, B)
        enc_at = dec_at = None
        if want_attn:
            # one probability tensor [B·nW, heads, N, N] per stage (its last block: reference model.py:859-860, 959-960), collected in
            # execution order: the encoder's stages, then the decoder's
            sink, nl = self._engine.attn_sink, len(cfg.depths)
            enc_at, dec_at = tuple(sink[:nl]), tuple(sink[nl:])
            self._engine.attn_sink = []
        if not return_dict:
            # reference model.py:1486-1488: (prediction,) + decoder_output[1:] + encoder_outputs[1:].  In tuple mode a stage
            # stack returns (last, all_hidden_states[, attentions]) without the reshaped copies (model.py:1087-1092, 1228-1233);
            # the decoder gets the caller's output_hidden_states, the ENCODER always True (model.py:1371-1378), so the
            # encoder's hidden states are always present
            out = (pred,) + ((dec_hs,) if want_hs else ()) + ((dec_at,) if want_attn else ()) + (enc_hs,) + ((enc_at,) if want_attn else ())
            return ((loss,) + out) if loss is not None else out
        if not want_hs:
            hs = rhs = None
        return ScOTOutput(loss=loss, output=pred, hidden_states=hs, attentions=(dec_at + enc_at) if want_attn else None,
                          reshaped_hidden_states=rhs)

    def _forward_resized(self, pv, t, lab, pixel_mask, in_size, want_grad):
        cfg = self.config
        if want_grad:
            _, pred = _ScOTFunction.apply(self._anchor, self, pv, t, None, None)
        else:
            _, pred, _ = self._engine.forward(pv, t, None, None, train=False, stochastic=self.training)
        pred = self._upsample(pred, in_size) if in_size > cfg.image_size else self._downsample(pred, in_size)
        if pixel_mask is not None:
            m = pixel_mask.view(pixel_mask.shape[0], pixel_mask.shape[1], 1, 1).expand_as(pred) if pixel_mask.dim() == 2 else pixel_mask
            pred = torch.where(m, lab, pred)
        loss = None
        if lab is not None:
            loss = _torch_loss(pred, lab, cfg.p, cfg.channel_slice_list_normalized_loss)
        return loss, pred

    def _hidden_tuples(self, hd, he, B):
        def shp(lst, ress):
            flat, resh = [], []
            for h, (hh, ww) in zip(lst, ress):
                c = h.shape[-1]
                f = h.view(B, hh * ww, c)
                flat.append(f)
                resh.append(f.view(B, hh, ww, c).permute(0, 3, 1, 2))
            return flat, resh
        _, enc, dec = stage_plan(self.config)
        enc_res = [enc[0].res] + [s.res for s in enc]
        dec_res = [dec[0].res] + [s.res for s in dec]
        fd, rd = shp(hd, dec_res)
        fe, re_ = shp(he, enc_res)
        return tuple(fd + fe), tuple(rd + re_), tuple(fe), tuple(re_), tuple(fd), tuple(rd)


# ---------------------------------------------------------------------------------------------- spectral resize
_RESIZE_OPS = {}


def resize_operator(s: int, t: int):
    """The reference resamples square images by  fft2 -> crop (model.py:1293-1300) / centred zero-pad (model.py:1302-1316) of the
    spectrum -> ifft2 -> real part, with norm="forward".  For a real image that is the linear map  Y = Re(P X P^T)  with

        P[m, n] = (1/s) Σ_{k=-q/2}^{q/2-1} exp(2 pi i k (m/t - n/s)),      q = min(s, t)        (t x s, complex)

    (down: the kept frequencies are fftfreq in [-t/2, t/2-1]; up: the whole source band [-s/2, s/2-1] is embedded in the centre of
    the target spectrum).  Returns (Re P, Im P) as float64 arrays."""
    import numpy as np
    q = min(s, t)
    k = np.arange(-(q // 2), q - q // 2, dtype=np.float64)
    if q % 2:   # fftfreq of an odd size is symmetric: -(q-1)/2 .. (q-1)/2
        k = np.arange(-(q - 1) // 2, (q - 1) // 2 + 1, dtype=np.float64)
    m = np.arange(t, dtype=np.float64)[:, None, None] / t
    n = np.arange(s, dtype=np.float64)[None, :, None] / s
    ph = 2.0 * np.pi * k[None, None, :] * (m - n)
    return np.cos(ph).sum(-1) / s, np.sin(ph).sum(-1) / s


def _resize_mats(s: int, t: int, device):
    key = (s, t, str(device))
    m = _RESIZE_OPS.get(key)
    if m is None:
        pr, pi = resize_operator(s, t)
        f = lambda a: torch.as_tensor(a, dtype=torch.float32).contiguous().to(device)
        # forward: U = X [Pr; Pi]^T then Y = Pr U_r - Pi U_i;  backward (dX = Pr^T dY Pr - Pi^T dY Pi): the same with P^T
        m = dict(cat=f(__import__("numpy").concatenate([pr, pi], 0)), pr=f(pr), pi=f(pi),
                 catT=f(__import__("numpy").concatenate([pr.T, pi.T], 0)), prT=f(pr.T), piT=f(pi.T))
        _RESIZE_OPS[key] = m
    return m


def _resize_apply(x, cat, pr, pi, s, t):
    from poseidon_amd import ops
    nimg = x.numel() // (s * s)
    U = torch.empty(nimg * s, 2 * t, dtype=torch.float32, device=x.device)
    ops.linear_fwd(ops.F32, x.reshape(nimg * s, s), cat, U)          # exact fp32 MFMA
    Y = torch.empty(nimg, t, t, dtype=torch.float32, device=x.device)
    ops.spectral_apply(U, pr, pi, Y, nimg, s, t)
    return Y


class _SpectralResize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, target):
        s = image.shape[-1]
        if image.shape[-2] != s:
            raise ValueError("spectral resize assumes square images (as the reference does)")
        m = _resize_mats(s, target, image.device)
        ctx.m, ctx.s, ctx.t = m, s, target
        x = image.to(torch.float32).contiguous()
        return _resize_apply(x, m["cat"], m["pr"], m["pi"], s, target).view(*image.shape[:-2], target, target)

    @staticmethod
    def backward(ctx, g):
        m, s, t = ctx.m, ctx.s, ctx.t
        dx = _resize_apply(g.to(torch.float32).contiguous(), m["catT"], m["prT"], m["piT"], t, s)
        return dx.view(*g.shape[:-2], s, s), None


def spectral_resize(image: torch.Tensor, target_size: int) -> torch.Tensor:
    """reference ScOT._downsample / _upsample (model.py:1293-1316) as two launches of the native library: an NT GEMM on the exact
    fp32 MFMA and scot_spectral_apply; differentiable (the adjoint is the same pair with the transposed operator)."""
    if not image.is_cuda:
        raise ScotLibraryError("spectral_resize needs CUDA(HIP) tensors: no CPU path exists in the product")
    if image.shape[-1] == target_size:
        return image
    return _SpectralResize.apply(image, int(target_size))


def _torch_loss(pred, labels, p, groups):
    """Loss at a resolution different from config.image_size (spectral-resize path only; off the hot path)."""
    fn = (lambda a, b: (a - b).abs().mean()) if p == 1 else (lambda a, b: ((a - b) ** 2).mean())
    if groups is None:
        return fn(pred, labels)
    terms = [fn(pred[:, groups[i]:groups[i + 1]], labels[:, groups[i]:groups[i + 1]]) /
             (fn(labels[:, groups[i]:groups[i + 1]], torch.zeros_like(labels[:, groups[i]:groups[i + 1]])) + 1e-10)
             for i in range(len(groups) - 1)]
    return torch.stack(terms).mean()
