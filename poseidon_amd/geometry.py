"""Host-side geometry of the scOT hot path: token grids, clamped windows/shifts, the state-dict schema.

Everything here is integer bookkeeping that the reference spreads over module constructors
(reference scOT/model.py:385-440 window/shift clamp, :790-794 / :885-902 block shift order,
:1024-1073 / :1118-1141 stage resolutions) — restated once so that the engine, the nn.Module mirror and the
tests agree on it.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Tuple


def clamp_window_shift(res: int, target_window: int, target_shift: int) -> Tuple[int, int]:
    """reference model.py:412-440."""
    window = res if res <= target_window else target_window
    shift = 0 if res <= window else target_shift
    return window, shift


@dataclass
class BlockGeom:
    prefix: str          # state-dict prefix, e.g. "encoder.layers.0.blocks.1"
    dim: int
    heads: int
    res: Tuple[int, int]  # token grid at run time
    table_window: int    # constructor-time clamped window (the CPB table size; reference model.py:385-396)
    target_shift: int    # constructor-time clamped shift (sticky: reference model.py:412-440 is called twice)

    def window_shift(self) -> Tuple[int, int]:
        """Run-time window/shift for this block's grid (second, per-forward clamp)."""
        return clamp_window_shift(self.res[0], self.table_window, self.target_shift)


@dataclass
class StageGeom:
    prefix: str
    dim: int
    heads: int
    res: Tuple[int, int]
    blocks: List[BlockGeom]
    resample: str        # "downsample" | "upsample" | ""
    out_res: Tuple[int, int]


def stage_plan(cfg) -> Tuple[Tuple[int, int], List[StageGeom], List[StageGeom]]:
    """Returns (patch grid, encoder stages, decoder stages in execution order = deep→shallow)."""
    patch = cfg.patch_size
    gh = gw = cfg.image_size // patch
    nl = len(cfg.depths)
    enc: List[StageGeom] = []
    h, w = gh, gw
    for s in range(nl):
        dim = int(cfg.embed_dim * 2 ** s)
        ctor_res = gh // (2 ** s)
        blocks = [BlockGeom(f"encoder.layers.{s}.blocks.{i}", dim, cfg.num_heads[s], (h, w),
                            *clamp_window_shift(ctor_res, cfg.window_size, 0 if i % 2 == 0 else cfg.window_size // 2))
                  for i in range(cfg.depths[s])]
        last = s == nl - 1
        out = (h, w) if last else ((h + 1) // 2, (w + 1) // 2)
        enc.append(StageGeom(f"encoder.layers.{s}", dim, cfg.num_heads[s], (h, w), blocks,
                             "" if last else "downsample", out))
        h, w = out
    dec: List[StageGeom] = []
    hh, ww = enc[-1].res
    for k in range(nl):
        i_layer = nl - 1 - k
        dim = int(cfg.embed_dim * 2 ** i_layer)
        ctor_res = gh // (2 ** i_layer)
        depth = cfg.depths[i_layer]
        blocks = []
        for j in range(depth):
            i = depth - 1 - j  # reversed construction order (reference model.py:885-902)
            blocks.append(BlockGeom(f"decoder.layers.{k}.blocks.{j}", dim, cfg.num_heads[i_layer], (hh, ww),
                                    *clamp_window_shift(ctor_res, cfg.window_size,
                                                        0 if i % 2 == 0 else cfg.window_size // 2)))
        if i_layer > 0:
            out = (gh // (2 ** (i_layer - 1)), gw // (2 ** (i_layer - 1)))
        else:
            out = (hh, ww)
        dec.append(StageGeom(f"decoder.layers.{k}", dim, cfg.num_heads[i_layer], (hh, ww), blocks,
                             "upsample" if i_layer > 0 else "", out))
        hh, ww = out
    return (gh, gw), enc, dec


def drop_path_rates(cfg) -> Dict[str, float]:
    """Stochastic-depth rate of every ScOTLayer, keyed by its state_dict prefix.

    Reference: one `linspace(0, drop_path_rate, 2·Σdepths)`; the encoder takes the first half in block order
    (model.py:976-996, block i of a stage gets `drop_path[i]`, :795); the decoder takes the second half, stage of
    resolution level i_layer gets `dpr[Σ depths[i_layer+1:] : Σ depths[i_layer:]]` (:1111-1131) and — its blocks being
    built in reversed order with `drop_path[depth-1-i]` (:885-899) — block at POSITION j of the stage gets entry j."""
    depths = list(cfg.depths)
    total = sum(depths)
    n = 2 * total
    rate = float(cfg.drop_path_rate)
    lin = [rate * i / (n - 1) if n > 1 else 0.0 for i in range(n)]
    out: Dict[str, float] = {}
    for s, d in enumerate(depths):
        for i in range(d):
            out[f"encoder.layers.{s}.blocks.{i}"] = lin[sum(depths[:s]) + i]
    dec = lin[total:]
    nl = len(depths)
    for k in range(nl):
        i_layer = nl - 1 - k
        sl = dec[sum(depths[i_layer + 1:]): sum(depths[i_layer:])]
        for j in range(depths[i_layer]):
            out[f"decoder.layers.{k}.blocks.{j}"] = sl[j]
    return out


def _norm_keys(out: "OrderedDict[str, Tuple[int, ...]]", prefix: str, dim: int, cond: bool) -> None:
    if cond:
        out[prefix + ".weight.weight"] = (dim, 1)
        out[prefix + ".weight.bias"] = (dim,)
        out[prefix + ".bias.weight"] = (dim, 1)
        out[prefix + ".bias.bias"] = (dim,)
    else:
        out[prefix + ".weight"] = (dim,)
        out[prefix + ".bias"] = (dim,)


def _block_keys(out, prefix: str, dim: int, heads: int, cfg) -> None:
    a = prefix + ".attention.self."
    out[a + "logit_scale"] = (heads, 1, 1)
    out[a + "continuous_position_bias_mlp.0.weight"] = (512, 2)
    out[a + "continuous_position_bias_mlp.0.bias"] = (512,)
    out[a + "continuous_position_bias_mlp.2.weight"] = (heads, 512)
    out[a + "query.weight"] = (dim, dim)
    if cfg.qkv_bias:
        out[a + "query.bias"] = (dim,)
    out[a + "key.weight"] = (dim, dim)
    out[a + "value.weight"] = (dim, dim)
    if cfg.qkv_bias:
        out[a + "value.bias"] = (dim,)
    out[prefix + ".attention.output.dense.weight"] = (dim, dim)
    out[prefix + ".attention.output.dense.bias"] = (dim,)
    _norm_keys(out, prefix + ".layernorm_before", dim, cfg.use_conditioning)
    hid = int(cfg.mlp_ratio * dim)
    out[prefix + ".intermediate.dense.weight"] = (hid, dim)
    out[prefix + ".intermediate.dense.bias"] = (hid,)
    out[prefix + ".output.dense.weight"] = (dim, hid)
    out[prefix + ".output.dense.bias"] = (dim,)
    _norm_keys(out, prefix + ".layernorm_after", dim, cfg.use_conditioning)


def param_shapes(cfg, use_mask_token: bool = False) -> "OrderedDict[str, Tuple[int, ...]]":
    """State-dict key → shape, in the reference's registration order."""
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    c0, p = cfg.embed_dim, cfg.patch_size
    cond = cfg.use_conditioning
    if use_mask_token:
        out["embeddings.mask_token"] = (1, 1, c0)
    if cfg.use_absolute_embeddings:
        out["embeddings.position_embeddings"] = (1, (cfg.image_size // p) ** 2, c0)
    out["embeddings.patch_embeddings.projection.weight"] = (c0, cfg.num_channels, p, p)
    out["embeddings.patch_embeddings.projection.bias"] = (c0,)
    _norm_keys(out, "embeddings.norm", c0, cond)
    _, enc, dec = stage_plan(cfg)
    for st in enc:
        for b in st.blocks:
            _block_keys(out, b.prefix, b.dim, b.heads, cfg)
        if st.resample:
            out[st.prefix + ".downsample.reduction.weight"] = (2 * st.dim, 4 * st.dim)
            _norm_keys(out, st.prefix + ".downsample.norm", 2 * st.dim, cond)
    for st in dec:
        for b in st.blocks:
            _block_keys(out, b.prefix, b.dim, b.heads, cfg)
        if st.resample:
            out[st.prefix + ".upsample.upsample.weight"] = (2 * st.dim, st.dim)
            out[st.prefix + ".upsample.mixup.weight"] = (st.dim // 2, st.dim // 2)
            _norm_keys(out, st.prefix + ".upsample.norm", st.dim // 2, cond)
    out["patch_recovery.projection.weight"] = (c0, cfg.num_out_channels, p, p)
    out["patch_recovery.projection.bias"] = (cfg.num_out_channels,)
    out["patch_recovery.mixup.weight"] = (cfg.num_out_channels, cfg.num_out_channels, 5, 5)
    if cfg.residual_model != "convnext":
        raise ValueError("only residual_model='convnext' is on the hot path (SURVEY.md §8a row 18)")
    for i, depth in enumerate(cfg.skip_connections):
        dim = cfg.embed_dim * 2 ** i
        for j in range(int(depth)):
            pre = f"residual_blocks.{i}.{j}"
            out[pre + ".weight"] = (dim,)
            out[pre + ".dwconv.weight"] = (dim, 1, 7, 7)
            out[pre + ".dwconv.bias"] = (dim,)
            _norm_keys(out, pre + ".norm", dim, cond)
            out[pre + ".pwconv1.weight"] = (4 * dim, dim)
            out[pre + ".pwconv1.bias"] = (4 * dim,)
            out[pre + ".pwconv2.weight"] = (dim, 4 * dim)
            out[pre + ".pwconv2.bias"] = (dim,)
    return out


def count_params(cfg) -> int:
    return sum(math.prod(s) for s in param_shapes(cfg).values())
