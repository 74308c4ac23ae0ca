"""ScOTConfig — hyper-parameters of the scOT hot path, JSON-compatible with the reference's HF config.

Mirrors the fields/defaults of the reference `ScOTConfig` (reference scOT/model.py:66-132) and the
`config.json` key set of SURVEY.md A.3, without depending on `transformers`.
"""
from __future__ import annotations

import copy
import json
import os
from typing import Any, Dict, List, Optional

MODEL_MAP = {  # presets of reference scOT/train.py:35-72 (values only)
    "T": dict(num_heads=[3, 6, 12, 24], skip_connections=[2, 2, 2, 0], window_size=16, patch_size=4,
              mlp_ratio=4.0, depths=[4, 4, 4, 4], embed_dim=48),
    "S": dict(num_heads=[3, 6, 12, 24], skip_connections=[2, 2, 2, 0], window_size=16, patch_size=4,
              mlp_ratio=4.0, depths=[8, 8, 8, 8], embed_dim=48),
    "B": dict(num_heads=[3, 6, 12, 24], skip_connections=[2, 2, 2, 0], window_size=16, patch_size=4,
              mlp_ratio=4.0, depths=[8, 8, 8, 8], embed_dim=96),
    "L": dict(num_heads=[3, 6, 12, 24], skip_connections=[2, 2, 2, 0], window_size=16, patch_size=4,
              mlp_ratio=4.0, depths=[8, 8, 8, 8], embed_dim=192),
}


class ScOTConfig:
    model_type = "swinv2"  # reference model.py:69
    attribute_map = {"num_attention_heads": "num_heads", "num_hidden_layers": "num_layers"}

    def __init__(self, image_size=224, patch_size=4, num_channels=3, num_out_channels=1, embed_dim=96,
                 depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), skip_connections=(True, True, True),
                 window_size=7, mlp_ratio=4.0, qkv_bias=True, hidden_dropout_prob=0.0,
                 attention_probs_dropout_prob=0.0, drop_path_rate=0.1, hidden_act="gelu",
                 use_absolute_embeddings=False, initializer_range=0.02, layer_norm_eps=1e-5, p=1,
                 channel_slice_list_normalized_loss=None, residual_model="convnext", use_conditioning=False,
                 learn_residual=False, **kwargs):
        self.image_size = image_size
        self.patch_size = patch_size
        self.num_channels = num_channels
        self.embed_dim = embed_dim
        self.depths = list(depths)
        self.num_layers = len(self.depths)
        self.num_heads = list(num_heads)
        self.skip_connections = list(skip_connections)
        self.window_size = window_size
        self.mlp_ratio = mlp_ratio
        self.qkv_bias = qkv_bias
        self.hidden_dropout_prob = hidden_dropout_prob
        self.attention_probs_dropout_prob = attention_probs_dropout_prob
        self.drop_path_rate = drop_path_rate
        self.hidden_act = hidden_act
        self.use_absolute_embeddings = use_absolute_embeddings
        self.use_conditioning = use_conditioning
        self.learn_residual = learn_residual if self.use_conditioning else False  # reference model.py:122
        self.layer_norm_eps = layer_norm_eps
        self.initializer_range = initializer_range
        self.hidden_size = int(embed_dim * 2 ** (len(self.depths) - 1))
        self.pretrained_window_sizes = (0, 0, 0, 0)
        self.num_out_channels = num_out_channels
        self.p = p
        self.channel_slice_list_normalized_loss = channel_slice_list_normalized_loss
        self.residual_model = residual_model
        # HF PretrainedConfig attributes read by the reference harness / forward
        self.output_attentions = kwargs.pop("output_attentions", False)
        self.output_hidden_states = kwargs.pop("output_hidden_states", False)
        self.use_return_dict = kwargs.pop("return_dict", True)
        self.chunk_size_feed_forward = kwargs.pop("chunk_size_feed_forward", 0)
        for drop in ("model_type", "num_layers", "hidden_size", "pretrained_window_sizes", "transformers_version",
                     "architectures", "dtype", "torch_dtype"):
            kwargs.pop(drop, None)
        self._extra = dict(kwargs)

    # HF-style aliases
    @property
    def num_attention_heads(self):
        return self.num_heads

    @property
    def num_hidden_layers(self):
        return self.num_layers

    def to_dict(self) -> Dict[str, Any]:
        keys = ["attention_probs_dropout_prob", "channel_slice_list_normalized_loss", "depths", "drop_path_rate",
                "embed_dim", "hidden_act", "hidden_dropout_prob", "hidden_size", "image_size", "initializer_range",
                "layer_norm_eps", "learn_residual", "mlp_ratio", "num_channels", "num_heads", "num_layers",
                "num_out_channels", "p", "patch_size", "pretrained_window_sizes", "qkv_bias", "residual_model",
                "skip_connections", "use_absolute_embeddings", "use_conditioning", "window_size"]
        d = {k: copy.deepcopy(getattr(self, k)) for k in keys}
        d["pretrained_window_sizes"] = list(d["pretrained_window_sizes"])
        d["model_type"] = self.model_type
        d["architectures"] = ["ScOT"]
        d["dtype"] = "float32"
        d["transformers_version"] = "4.29.2"
        return d

    def to_json_string(self) -> str:
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"

    def save_pretrained(self, directory: str) -> None:
        os.makedirs(directory, exist_ok=True)
        with open(os.path.join(directory, "config.json"), "w") as f:
            f.write(self.to_json_string())

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "ScOTConfig":
        return cls(**copy.deepcopy(d))

    @classmethod
    def from_pretrained(cls, path: str, **overrides) -> "ScOTConfig":
        cfg_path = os.path.join(path, "config.json") if os.path.isdir(path) else path
        with open(cfg_path) as f:
            d = json.load(f)
        d.update(overrides)
        return cls.from_dict(d)

    def __repr__(self):
        return f"ScOTConfig {self.to_json_string()}"


def preset(size: str, **kw) -> ScOTConfig:
    """`ScOTConfig(...)` exactly as reference train.py:247-275 builds it from MODEL_MAP."""
    base = dict(qkv_bias=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, drop_path_rate=0.0,
                hidden_act="gelu", use_absolute_embeddings=False, initializer_range=0.02, layer_norm_eps=1e-5,
                p=1, residual_model="convnext", use_conditioning=True, learn_residual=False)
    base.update(MODEL_MAP[size])
    base.update(kw)
    return ScOTConfig(**base)
