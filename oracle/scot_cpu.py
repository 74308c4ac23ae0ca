"""ORACLE — CPU restatement of the reference's ScOT forward (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (poseidon_amd/, scOT/) never does and fails loudly when the HIP library is missing.

What it restates (reference = /root/reference/scOT/model.py, HF = transformers swinv2/modeling_swinv2.py,
the un-vendored dependency pinned ==4.29.2 in pyproject.toml:9; installed here: 5.15.0):
  * ConditionalLayerNorm / LayerNorm ........ model.py:135-160
  * ScOTPatchEmbeddings / ScOTEmbeddings .... model.py:249-366
  * ScOTLayer (shift/window clamp, pad, roll+partition, mask, res-post-norm) model.py:369-581
  * Swinv2SelfAttention (cosine attention, log-CPB bias, mask applied TWICE in 5.15.0) HF:359-492
  * Swinv2SelfOutput / Intermediate / Output  HF:496-561
  * ScOTPatchMerging / Unmerging / Recovery .. model.py:584-760
  * encode/decode stages and skip wiring ..... model.py:763-1240
  * ConvNeXtBlock ........................... model.py:163-217
  * ScOT.forward incl. learn_residual, pixel_mask overwrite and the grouped relative loss  model.py:1318-1509
It is written with explicit gather indices / analytic masks instead of roll/partition so that it is an
independent restatement, uses only torch CPU ops (no `transformers`), and is differentiable through
torch autograd so that it is also the backward oracle.

PINNING: the reference has no tests or golden vectors → "parity unpinned" by the
reference itself.  This oracle is pinned against fixtures generated in the build container by importing
the real reference (tests/golden/make_fixtures.py, committed) — see tests/test_oracle_golden.py.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------- geometry
def clamp_window_shift(res: int, target_window: int, target_shift: int) -> Tuple[int, int]:
    """model.py:412-440 — window=min(res,target); shift=0 if res<=window else target_shift.
    NB the reference calls this TWICE with sticky state: once in the constructor with the constructor-time
    resolution grid//2^stage (model.py:385) and again every forward with the run-time grid (model.py:509-510),
    the second call seeing the already-clamped window/shift.  `ctor_window_shift` models the first call."""
    window = res if res <= target_window else target_window
    shift = 0 if res <= window else target_shift
    return window, shift


def ctor_window_shift(ctor_res: int, cfg_window: int, block_target_shift: int) -> Tuple[int, int]:
    return clamp_window_shift(ctor_res, cfg_window, block_target_shift)


def window_gather_index(hp: int, wp: int, ws: int, shift: int) -> Tensor:
    """roll(-s,-s)+window_partition == gather of token ((y+s)%H,(x+s)%W)  (SURVEY §8a row 10).
    Returns LongTensor [nW, ws*ws] of flat indices into the (hp*wp) padded grid."""
    nwy, nwx = hp // ws, wp // ws
    wy = torch.arange(nwy).view(nwy, 1, 1, 1)
    wx = torch.arange(nwx).view(1, nwx, 1, 1)
    iy = torch.arange(ws).view(1, 1, ws, 1)
    ix = torch.arange(ws).view(1, 1, 1, ws)
    y = (wy * ws + iy + shift) % hp
    x = (wx * ws + ix + shift) % wp
    return (y * wp + x).reshape(nwy * nwx, ws * ws)


def shift_mask(hp: int, wp: int, ws: int, shift: int, value: float = -100.0) -> Optional[Tensor]:
    """model.py:442-478 analytic form: region id of a *shifted-grid* coordinate n along an axis of
    length L is (n>=L-ws)+(n>=L-shift); pairs with different (ry,rx) get `value`."""
    if shift == 0:
        return None
    nwy, nwx = hp // ws, wp // ws
    ys = torch.arange(hp)
    xs = torch.arange(wp)
    ry = (ys >= hp - ws).long() + (ys >= hp - shift).long()
    rx = (xs >= wp - ws).long() + (xs >= wp - shift).long()
    rid = (ry.view(hp, 1) * 3 + rx.view(1, wp))  # on the shifted grid
    rid = rid.view(nwy, ws, nwx, ws).permute(0, 2, 1, 3).reshape(nwy * nwx, ws * ws)
    diff = rid.unsqueeze(1) != rid.unsqueeze(2)
    return diff.to(torch.float32) * value


def cpb_coords_table(ws: int) -> Tensor:
    """HF:457-476: sign(d)*log2(1+8|d|/(ws-1))/3 for d in [-(ws-1), ws-1]^2 → [(2ws-1)^2, 2]."""
    r = torch.arange(-(ws - 1), ws, dtype=torch.float32)
    tab = torch.stack(torch.meshgrid(r, r, indexing="ij"), dim=-1)  # [2ws-1, 2ws-1, 2] (dy, dx)
    if ws > 1:
        tab = tab / (ws - 1)
    tab = tab * 8
    tab = torch.sign(tab) * torch.log2(torch.abs(tab) + 1.0) / math.log2(8)
    return tab.reshape(-1, 2)


def cpb_index(ws: int) -> Tensor:
    """HF:481-490: index[i,j] = (yi-yj+ws-1)*(2ws-1) + (xi-xj+ws-1)."""
    yy = torch.arange(ws).repeat_interleave(ws)
    xx = torch.arange(ws).repeat(ws)
    dy = yy.view(-1, 1) - yy.view(1, -1) + ws - 1
    dx = xx.view(-1, 1) - xx.view(1, -1) + ws - 1
    return dy * (2 * ws - 1) + dx


# ----------------------------------------------------------------------------- small ops
def norm(sd: Dict[str, Tensor], prefix: str, x: Tensor, time: Optional[Tensor], eps: float, cond: bool) -> Tensor:
    """ConditionalLayerNorm (model.py:143-160) or LayerNorm-ignoring-time (model.py:135-140)."""
    if not cond:
        return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)
    mean = x.mean(dim=-1, keepdim=True)
    var = (x * x).mean(dim=-1, keepdim=True) - mean * mean  # biased, no clamp (model.py:152)
    xh = (x - mean) / torch.sqrt(var + eps)
    t = time.reshape(-1, 1).to(x.dtype)
    g = t @ sd[prefix + ".weight.weight"].t() + sd[prefix + ".weight.bias"]  # [B, C]
    b = t @ sd[prefix + ".bias.weight"].t() + sd[prefix + ".bias.bias"]
    shape = [x.shape[0]] + [1] * (x.dim() - 2) + [x.shape[-1]]
    return g.view(shape) * xh + b.view(shape)


def gelu(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x * 0.7071067811865476))


def attention(sd, prefix: str, xw: Tensor, heads: int, ws: int, mask: Optional[Tensor], attn_sink: Optional[list] = None) -> Tensor:
    """Swinv2SelfAttention + SelfOutput on windows xw [Bw, N, C] (HF:389-455, 502-506).  attn_sink: receives the attention
    probabilities [Bw, heads, N, N] (what `output_attentions=True` returns, HF:443-455)."""
    bw, n, c = xw.shape
    d = c // heads
    p = prefix + ".self."
    q = xw @ sd[p + "query.weight"].t()
    if (p + "query.bias") in sd:
        q = q + sd[p + "query.bias"]
    k = xw @ sd[p + "key.weight"].t()  # key has NO bias (HF:385)
    v = xw @ sd[p + "value.weight"].t()
    if (p + "value.bias") in sd:
        v = v + sd[p + "value.bias"]
    q = q.view(bw, n, heads, d).transpose(1, 2)
    k = k.view(bw, n, heads, d).transpose(1, 2)
    v = v.view(bw, n, heads, d).transpose(1, 2)
    qn = q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12)  # F.normalize eps
    kn = k / k.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    scale = torch.exp(torch.clamp(sd[p + "logit_scale"], max=math.log(100.0)))  # [h,1,1]
    s = (qn @ kn.transpose(-1, -2)) * scale
    tab = cpb_coords_table(ws).to(xw.dtype)
    hid = torch.relu(tab @ sd[p + "continuous_position_bias_mlp.0.weight"].t() + sd[p + "continuous_position_bias_mlp.0.bias"])
    tbl = hid @ sd[p + "continuous_position_bias_mlp.2.weight"].t()  # [(2ws-1)^2, heads]
    bias = 16.0 * torch.sigmoid(tbl[cpb_index(ws).reshape(-1)].view(n, n, heads).permute(2, 0, 1))
    s = s + bias.unsqueeze(0)
    if mask is not None:
        nw = mask.shape[0]
        s = s.view(bw // nw, nw, heads, n, n) + 2.0 * mask.view(1, nw, 1, n, n)  # added twice (HF:433-436)
        s = s.view(bw, heads, n, n)
    pr = torch.softmax(s, dim=-1)
    if attn_sink is not None:
        attn_sink.append(pr)
    o = (pr @ v).transpose(1, 2).reshape(bw, n, c)
    return o @ sd[prefix + ".output.dense.weight"].t() + sd[prefix + ".output.dense.bias"]


def scot_layer(sd, prefix: str, x: Tensor, hw: Tuple[int, int], time, heads: int, target_window: int,
               target_shift: int, eps: float, cond: bool, drop_masks=None, attn_sink: Optional[list] = None) -> Tensor:
    """ScOTLayer.forward (model.py:500-581) — res-post-norm.  drop_masks: {(prefix, 0|1): [B] mask/keep_prob} =
    Swinv2DropPath (HF:565-586) on the two normed branches (model.py:570,574) with the random draw supplied."""
    h, w = hw
    b, l, c = x.shape
    ws, shift = clamp_window_shift(h, target_window, target_shift)
    pad_r = (ws - w % ws) % ws
    pad_b = (ws - h % ws) % ws
    xg = x.view(b, h, w, c)
    if pad_r or pad_b:
        xg = F.pad(xg, (0, 0, 0, pad_r, 0, pad_b))
    hp, wp = h + pad_b, w + pad_r
    idx = window_gather_index(hp, wp, ws, shift)  # [nW, N]
    nw, n = idx.shape
    xw = xg.reshape(b, hp * wp, c)[:, idx.reshape(-1)].reshape(b * nw, n, c)
    mask = shift_mask(hp, wp, ws, shift)
    aw = attention(sd, prefix + ".attention", xw, heads, ws, mask, attn_sink)
    out = torch.zeros(b, hp * wp, c, dtype=x.dtype)
    out[:, idx.reshape(-1)] = aw.reshape(b, nw * n, c)
    out = out.view(b, hp, wp, c)[:, :h, :w].reshape(b, l, c)
    def dp(t, which):
        m = drop_masks.get((prefix, which)) if drop_masks else None
        return t if m is None else t * m.to(t.dtype).view(-1, 1, 1)
    hid = x + dp(norm(sd, prefix + ".layernorm_before", out, time, eps, cond), 0)
    y = gelu(hid @ sd[prefix + ".intermediate.dense.weight"].t() + sd[prefix + ".intermediate.dense.bias"])
    y = y @ sd[prefix + ".output.dense.weight"].t() + sd[prefix + ".output.dense.bias"]
    return hid + dp(norm(sd, prefix + ".layernorm_after", y, time, eps, cond), 1)


def patch_embed(sd, x: Tensor, patch: int) -> Tuple[Tensor, Tuple[int, int]]:
    """ScOTPatchEmbeddings (model.py:286-310): right/bottom zero pad, conv k=s=patch as an unfold+matmul."""
    b, cin, h, w = x.shape
    ph, pw = (patch - h % patch) % patch, (patch - w % patch) % patch
    if ph or pw:
        x = F.pad(x, (0, pw, 0, ph))
    gh, gw = x.shape[2] // patch, x.shape[3] // patch
    cols = x.view(b, cin, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5).reshape(b, gh * gw, cin * patch * patch)
    wgt = sd["embeddings.patch_embeddings.projection.weight"]
    y = cols @ wgt.reshape(wgt.shape[0], -1).t() + sd["embeddings.patch_embeddings.projection.bias"]
    return y, (gh, gw)


def patch_merge(sd, prefix: str, x: Tensor, hw, time, eps, cond) -> Tensor:
    """ScOTPatchMerging (model.py:680-712)."""
    h, w = hw
    b, l, c = x.shape
    xg = x.view(b, h, w, c)
    if h % 2 or w % 2:
        xg = F.pad(xg, (0, 0, 0, w % 2, 0, h % 2))
    cat = torch.cat([xg[:, 0::2, 0::2], xg[:, 1::2, 0::2], xg[:, 0::2, 1::2], xg[:, 1::2, 1::2]], dim=-1)
    cat = cat.reshape(b, -1, 4 * c)
    return norm(sd, prefix + ".norm", cat @ sd[prefix + ".reduction.weight"].t(), time, eps, cond)


def patch_unmerge(sd, prefix: str, x: Tensor, out_hw, time, eps, cond) -> Tensor:
    """ScOTPatchUnmerging (model.py:737-760). NB: its norm is built with the DEFAULT eps (1e-5):
    `norm_layer(dim // 2)` (model.py:733), same for merging (model.py:670) and embeddings (model.py:342)."""
    b, l, c = x.shape
    hi = int(math.floor(l ** 0.5))
    y = x @ sd[prefix + ".upsample.weight"].t()  # [b, l, 2c]
    y = y.view(b, hi, hi, 2, 2, c // 2).permute(0, 1, 3, 2, 4, 5).reshape(b, 2 * hi, 2 * hi, c // 2)
    y = y[:, : out_hw[0], : out_hw[1]].reshape(b, -1, c // 2)
    y = norm(sd, prefix + ".norm", y, time, eps, cond)
    return y @ sd[prefix + ".mixup.weight"].t()


def convnext(sd, prefix: str, x: Tensor, time, eps, cond) -> Tensor:
    """ConvNeXtBlock (model.py:198-217)."""
    b, l, c = x.shape
    hw = int(math.floor(l ** 0.5))
    y = x.view(b, hw, hw, c).permute(0, 3, 1, 2)
    y = F.conv2d(y, sd[prefix + ".dwconv.weight"], sd[prefix + ".dwconv.bias"], padding=3, groups=c)
    y = y.permute(0, 2, 3, 1)
    y = norm(sd, prefix + ".norm", y, time, eps, cond)
    y = gelu(y @ sd[prefix + ".pwconv1.weight"].t() + sd[prefix + ".pwconv1.bias"])
    y = y @ sd[prefix + ".pwconv2.weight"].t() + sd[prefix + ".pwconv2.bias"]
    if (prefix + ".weight") in sd:
        y = sd[prefix + ".weight"] * y
    return x + y.reshape(b, l, c)


def patch_recovery(sd, x: Tensor, grid: Tuple[int, int], patch: int, image: Tuple[int, int]) -> Tensor:
    """ScOTPatchRecovery (model.py:639-647): ConvTranspose2d(k=s=patch) as matmul+pixel scatter, crop, 5x5 conv."""
    b, l, c = x.shape
    wgt = sd["patch_recovery.projection.weight"]  # (C, Cout, p, p)
    cout = wgt.shape[1]
    y = x @ wgt.reshape(c, cout * patch * patch)  # [b, l, cout*p*p]
    y = y.view(b, grid[0], grid[1], cout, patch, patch).permute(0, 3, 1, 4, 2, 5)
    y = y.reshape(b, cout, grid[0] * patch, grid[1] * patch) + sd["patch_recovery.projection.bias"].view(1, -1, 1, 1)
    y = y[:, :, : image[0], : image[1]]
    return F.conv2d(y, sd["patch_recovery.mixup.weight"], None, padding=2)


def scot_loss(pred: Tensor, labels: Tensor, p: int, groups: Optional[Sequence[int]]) -> Tensor:
    """model.py:1424-1484."""
    if p == 1:
        fn = lambda a, b_: (a - b_).abs().mean()
    elif p == 2:
        fn = lambda a, b_: ((a - b_) ** 2).mean()
    else:
        raise ValueError("p must be 1 or 2")
    if groups is None:
        return fn(pred, labels)
    terms = []
    for i in range(len(groups) - 1):
        a, b_ = groups[i], groups[i + 1]
        terms.append(fn(pred[:, a:b_], labels[:, a:b_]) / (fn(labels[:, a:b_], torch.zeros_like(labels[:, a:b_])) + 1e-10))
    return torch.stack(terms).mean()


# ----------------------------------------------------------------------------- whole model
def _get(cfg, name, default=None):
    return getattr(cfg, name, default) if not isinstance(cfg, dict) else cfg.get(name, default)


def spectral_resize(img: Tensor, target: int) -> Tensor:
    """model.py:1293-1316 (_downsample/_upsample), square images."""
    size = img.shape[-2]
    if target < size:
        freqs = torch.fft.fftfreq(size, d=1 / size)
        sel = torch.logical_and(freqs >= -target / 2, freqs <= target / 2 - 1)
        hat = torch.fft.fft2(img, norm="forward")[:, :, sel, :][:, :, :, sel]
        return torch.fft.ifft2(hat, norm="forward").real
    hat = torch.fft.fftshift(torch.fft.fft2(img, norm="forward"))
    pad = (target - size) // 2
    hat = torch.complex(F.pad(hat.real, (pad, pad, pad, pad)), F.pad(hat.imag, (pad, pad, pad, pad)))
    return torch.fft.ifft2(torch.fft.ifftshift(hat), norm="forward").real


def scot_forward(sd: Dict[str, Tensor], cfg, pixel_values: Tensor, time: Optional[Tensor] = None,
                 labels: Optional[Tensor] = None, pixel_mask: Optional[Tensor] = None,
                 return_intermediates: bool = False, drop_masks=None, bool_masked_pos: Optional[Tensor] = None,
                 output_attentions: bool = False):
    """ScOT.forward (model.py:1318-1509) → (loss or None, prediction[, intermediates]).  output_attentions: intermediates["attentions"]
    = the attention probabilities of every stage's LAST block, decoder stages first (model.py:859-860, 959-960, 1084-1085, 1225-1226,
    1497-1501) — implies return_intermediates."""
    if pixel_values is None:
        raise ValueError("pixel_values cannot be None")
    image_size = _get(cfg, "image_size")
    patch = _get(cfg, "patch_size", 4)
    depths = list(_get(cfg, "depths"))
    heads = list(_get(cfg, "num_heads"))
    window = _get(cfg, "window_size")
    eps = _get(cfg, "layer_norm_eps", 1e-5)
    cond = bool(_get(cfg, "use_conditioning", False))
    skips_cfg = list(_get(cfg, "skip_connections"))
    nl = len(depths)
    inter = {}
    enc_attn: List[Tensor] = []
    dec_attn: List[Tensor] = []
    return_intermediates = return_intermediates or output_attentions

    in_size = pixel_values.shape[2]
    if in_size != image_size:
        pixel_values = spectral_resize(pixel_values, image_size)

    x, (gh, gw) = patch_embed(sd, pixel_values, patch)
    x = norm(sd, "embeddings.norm", x, time, 1e-5, cond)  # default eps (model.py:342)
    if bool_masked_pos is not None:      # mask tokens (model.py:353-359)
        m = bool_masked_pos.reshape(x.shape[0], -1, 1).to(x.dtype)
        x = x * (1.0 - m) + sd["embeddings.mask_token"].expand(x.shape[0], x.shape[1], -1) * m
    if "embeddings.position_embeddings" in sd:
        x = x + sd["embeddings.position_embeddings"]
    inter["embeddings"] = x

    # encoder (model.py:816-861, 1008-1099)
    skip_states: List[Tensor] = []
    hw = (gh, gw)
    for s in range(nl):
        stage_in = x
        for i in range(depths[s]):
            cw, cs = ctor_window_shift(gh // (2 ** s), window, 0 if i % 2 == 0 else window // 2)
            x = scot_layer(sd, f"encoder.layers.{s}.blocks.{i}", x, hw, time, heads[s], cw, cs, eps, cond, drop_masks,
                           enc_attn if (output_attentions and i == depths[s] - 1) else None)
        skip_states.append(x)  # hidden_states_before_downsampling
        inter[f"enc{s}"] = x
        if s < nl - 1:
            x = patch_merge(sd, f"encoder.layers.{s}.downsample", x + stage_in, hw, time, 1e-5, cond)
            hw = ((hw[0] + 1) // 2, (hw[1] + 1) // 2)

    # ConvNeXt skip blocks (model.py:1388-1393)
    for i in range(len(skip_states)):
        nblk = int(skips_cfg[i]) if i < len(skips_cfg) else 0
        for j in range(nblk):
            skip_states[i] = convnext(sd, f"residual_blocks.{i}.{j}", skip_states[i], time, eps, cond)

    # decoder (model.py:916-961, 1145-1240): stages deep→shallow, module index k = nl-1-i_layer
    x = skip_states[-1]
    dim0 = int(math.floor(x.shape[1] ** 0.5))
    hw = (dim0, dim0)
    skips = skip_states[:-1]
    for k in range(nl):
        i_layer = nl - 1 - k
        if k != 0 and skips[len(skips) - k] is not None:
            x = x + skips[len(skips) - k]
        depth = depths[i_layer]
        for j in range(depth):
            i = depth - 1 - j  # blocks are built for i in reversed(range(depth)) (model.py:885-902)
            cw, cs = ctor_window_shift(gh // (2 ** i_layer), window, 0 if i % 2 == 0 else window // 2)
            x = scot_layer(sd, f"decoder.layers.{k}.blocks.{j}", x, hw, time, heads[i_layer], cw, cs, eps, cond, drop_masks,
                           dec_attn if (output_attentions and j == depth - 1) else None)
        inter[f"dec{k}"] = x
        if i_layer > 0:
            up = (gh // (2 ** (i_layer - 1)), gw // (2 ** (i_layer - 1)))
            x = patch_unmerge(sd, f"decoder.layers.{k}.upsample", x, up, time, 1e-5, cond)
            hw = up

    pred = patch_recovery(sd, x, (image_size // patch, image_size // patch), patch, (image_size, image_size))
    if bool(_get(cfg, "learn_residual", False)) and cond:
        nout = _get(cfg, "num_out_channels")
        pred = pred + pixel_values[:, :nout]
    if in_size != image_size:
        pred = spectral_resize(pred, in_size)
    if pixel_mask is not None:
        pred = torch.where(_broadcast_mask(pixel_mask, pred), labels.to(pred.dtype), pred)
    loss = None
    if labels is not None:
        loss = scot_loss(pred, labels, _get(cfg, "p", 1), _get(cfg, "channel_slice_list_normalized_loss"))
    if output_attentions:
        inter["attentions"] = tuple(dec_attn) + tuple(enc_attn)
    if return_intermediates:
        return loss, pred, inter
    return loss, pred


def _broadcast_mask(mask: Tensor, like: Tensor) -> Tensor:
    """`prediction[pixel_mask] = labels[pixel_mask]` (model.py:1422-1423) with a (B,C) bool mask selects whole
    (H,W) planes; a full-shape or broadcastable mask selects elements."""
    if mask.dim() == 2:
        return mask.view(mask.shape[0], mask.shape[1], 1, 1).expand_as(like)
    return mask.expand_as(like)
