"""Harness pins: optimizer group membership equals the reference's (fixture from the real
reference Trainer), on CPU.  Rollout numerics are GPU tests (tests/test_model_gpu.py)."""
import json
import os

import pytest
import torch

from conftest import GOLDEN
from poseidon_amd.config import preset
from poseidon_amd.harness import create_optimizer, optimizer_param_groups
from scOT.model import ScOT


def test_optimizer_groups_match_reference_trainer():
    ref = json.load(open(os.path.join(GOLDEN, "optimizer_groups_T.json")))
    cfg = preset("T", image_size=128, num_channels=4, num_out_channels=4, channel_slice_list_normalized_loss=[0, 1, 3, 4])
    model = ScOT(cfg)
    groups = optimizer_param_groups(model, weight_decay=1e-6, learning_rate_embedding_recovery=1e-4,
                                    learning_rate_time_embedding=2e-4, return_names=True)
    assert [len(g["names"]) for g in groups] == [257, 274, 9, 304]
    for gi, g in enumerate(groups):
        r = ref[f"group{gi}"]
        assert sorted(g["names"]) == r["names"]
        assert g["weight_decay"] == r["weight_decay"]
        if "lr" in g:
            assert g["lr"] == r["lr"]
    numel = [sum(p.numel() for p in g["params"]) for g in groups]
    assert numel == [20431272, 283584, 6788, 52800]
    # quirks
    names = {n: gi for gi, g in enumerate(groups) for n in g["names"]}
    assert names["encoder.layers.0.blocks.0.attention.self.continuous_position_bias_mlp.0.weight"] == 1
    assert names["encoder.layers.0.blocks.0.attention.self.logit_scale"] == 0
    assert names["residual_blocks.0.0.weight"] == 0
    assert names["embeddings.norm.weight.weight"] == 2
    # 2-group and 3-group variants
    g2 = optimizer_param_groups(model, weight_decay=0.1)
    assert len(g2) == 2 and sum(len(g["params"]) for g in g2) == 844
    g3 = optimizer_param_groups(model, weight_decay=0.1, learning_rate_time_embedding=1e-3)
    assert len(g3) == 3 and len(g3[2]["params"]) == 308  # all cond-LN tensors incl. embeddings.norm
    opt = create_optimizer(model, 5e-4, 1e-6, learning_rate_embedding_recovery=1e-4, learning_rate_time_embedding=2e-4)
    assert [g["lr"] for g in opt.param_groups] == [5e-4, 5e-4, 1e-4, 2e-4]


def test_fused_optimizer_group_map_covers_exactly_the_parameters():
    """poseidon_amd.optim.group_map8: every parameter's arena range carries its group id, everything else (alignment padding,
    the key-bias slot of the fused qkv bias) is marked 'not a parameter'."""
    import numpy as np
    from poseidon_amd.arena import Arena
    from poseidon_amd.config import preset
    from poseidon_amd.geometry import param_shapes
    from poseidon_amd.optim import SKIP, group_map8
    cfg = preset("T", image_size=128, num_channels=4, num_out_channels=4)
    shapes = param_shapes(cfg)
    arena = Arena(shapes, "cpu", requires_grad_arena=False)
    names = list(shapes)
    groups = [names[0::3], names[1::3], names[2::3]]
    m = group_map8(arena, groups)
    owner = np.full(arena.size, SKIP, dtype=np.int64)
    for gi, g in enumerate(groups):
        for n in g:
            o = arena.offsets[n]
            owner[o:o + arena.numel(n)] = gi
    got = np.repeat(m, 8).astype(np.int64)
    assert (got[owner != SKIP] == owner[owner != SKIP]).all()          # every parameter element carries its group
    extra = (got != SKIP) & (owner == SKIP)                             # chunk tails of tensors whose size is not 8k: padding
    assert int(extra.sum()) < 8 * len(names) and float(arena.data[torch.from_numpy(extra)].abs().max()) == 0.0
    # the fused qkv bias keeps a zero slot for the bias-free key projection: it must never be stepped
    pre = "encoder.layers.0.blocks.0.attention.self."
    o, c = arena.offsets[pre + "qkv_bias"], shapes[pre + "query.bias"][0]
    assert (m[(o + c) // 8:(o + 2 * c) // 8] == SKIP).all() and (m[o // 8:(o + c) // 8] != SKIP).all()


def test_metrics_match_reference_pins():
    """scOT.metrics (numpy in → numpy out, torch in → torch out on the tensor's device) against values produced by the
    reference's metrics module (tests/golden/make_metrics_pins.py), including the zero-target guard and the group statistics
    of the reference's inference driver."""
    import json
    import os
    import numpy as np
    import torch
    from poseidon_amd.synth import closed_form_tensor
    from scOT import metrics as M
    pins = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "metrics_pins.json")))
    pr = np.asarray(closed_form_tensor("metrics:pred", (6, 4, 16, 16), 1.0), dtype=np.float32)
    tg = np.asarray(closed_form_tensor("metrics:target", (6, 4, 16, 16), 1.0), dtype=np.float32)
    tg[3] = 0.0

    def close(a, b, tol=2e-6):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        return np.all(np.abs(a - b) <= tol * np.maximum(np.abs(b), 1e-30))

    for p in (1, 2):
        for args in ((pr, tg), (torch.from_numpy(pr), torch.from_numpy(tg))):
            is_t = isinstance(args[0], torch.Tensor)
            e = M.lp_error(*args, p=p)
            assert isinstance(e, torch.Tensor) == is_t and close(e, pins[f"lp_error_p{p}"])
            assert close(M.relative_lp_error(*args, p=p), pins[f"relative_lp_error_p{p}"])
            assert close(M.relative_lp_error(*args, p=p, return_percent=False), pins[f"relative_lp_error_p{p}_nopercent"])
            assert close(M.mean_relative_lp_error(*args, p=p), pins[f"mean_relative_p{p}"])
            assert close(M.median_relative_lp_error(*args, p=p), pins[f"median_relative_p{p}"])
    out = M.channel_group_metrics(torch.from_numpy(pr), torch.from_numpy(tg), [0, 1, 3, 4], ["rho", "uv", "p"])
    for i, n in enumerate(["rho", "uv", "p"]):
        g = pins[f"group{i}"]
        assert close(out[n + "/median_relative_l1_error"], g["median_rel"]) and close(out[n + "/mean_relative_l1_error"], g["mean_rel"])
        assert close(out[n + "/std_relative_l1_error"], g["std_rel"], 1e-5) and close(out[n + "/max_relative_l1_error"], g["max_rel"])
        assert close(out[n + "/median_l1_error"], g["median_abs"]) and close(out[n + "/std_l1_error"], g["std_abs"], 1e-5)
    assert close(out["mean_relative_l1_error"], np.mean([pins[f"group{i}"]["mean_rel"] for i in range(3)]))
    assert close(out["mean_over_median_l1_error"], np.mean([pins[f"group{i}"]["median_abs"] for i in range(3)]))
    single = M.channel_group_metrics(pr[:, :1], tg[:, :1], [0, 1], full_data=True)
    assert close(single["mean_relative_l1_error"], pins["group0"]["mean_rel"]) and len(single["full_data"]) == 6


def test_compute_loss_accepts_trainer_kwargs():
    """reference Trainer.compute_loss (trainer.py:605-635) as called by transformers >= 4.46 (extra `num_items_in_batch`): loss from a
    ModelOutput-like object or from tuple position 0; a missing loss raises the reference's ValueError."""
    from types import SimpleNamespace
    from scOT.trainer import compute_loss

    class Out(SimpleNamespace):
        def keys(self):
            return [k for k, v in vars(self).items() if v is not None]

    class M:
        config = SimpleNamespace(use_conditioning=True, num_channels=1, num_out_channels=1)

        def __init__(self, ret):
            self.ret, self.calls = ret, 0

        def __call__(self, **kw):
            self.calls += 1
            return self.ret
    x = dict(pixel_values=torch.zeros(1, 1, 2, 2), time=torch.ones(1), labels=torch.zeros(1, 1, 2, 2))
    m = M(Out(loss=torch.tensor(2.0), output=torch.zeros(1, 1, 2, 2)))
    assert float(compute_loss(m, x, num_items_in_batch=7)) == 2.0 and m.calls == 1
    loss, out = compute_loss(m, x, return_outputs=True, ar_steps=2)
    assert float(loss) == 2.0 and m.calls == 3 and out is m.ret
    assert float(compute_loss(M((torch.tensor(3.0), None)), dict(x), ar_steps=None)) == 3.0
    with pytest.raises(ValueError, match="did not return a loss"):
        compute_loss(M(Out(loss=None, output=torch.zeros(1, 1, 2, 2))), x)


def test_rollout_stacks_hidden_states_over_steps():
    """reference Trainer._model_forward with output_all_steps (trainer.py:472-479, 509-520 / 544-551, 584-595): the hidden states,
    reshaped hidden states and attentions of every AR step are collected and stacked on dim 1, layer by layer; the list mode
    appends each OUTPUT twice (the reference's duplicated append) but the hidden states once."""
    from types import SimpleNamespace
    from poseidon_amd.harness import rollout

    class M:
        config = SimpleNamespace(use_conditioning=True, num_channels=1, num_out_channels=1)

        def __init__(self):
            self.calls = 0

        def __call__(self, pixel_values, time, **kw):
            self.calls += 1
            k = float(self.calls)
            return SimpleNamespace(loss=torch.tensor(k), output=pixel_values + 1, hidden_states=(torch.full((2, 3), k), torch.full((2, 5), 10 * k)),
                                   reshaped_hidden_states=(torch.full((2, 1, 3), k),), attentions=None)
    x = dict(pixel_values=torch.zeros(2, 1, 2, 2), time=torch.ones(2))
    o = rollout(M(), x, 3, output_all_steps=True)
    assert o.output.shape == (2, 3, 1, 2, 2) and o.loss.tolist() == [1.0, 2.0, 3.0]
    assert [tuple(h.shape) for h in o.hidden_states] == [(2, 3, 3), (2, 3, 5)] and o.hidden_states[1][0, :, 0].tolist() == [10.0, 20.0, 30.0]
    assert tuple(o.reshaped_hidden_states[0].shape) == (2, 3, 1, 3) and o.attentions is None
    o = rollout(M(), x, [1, 2], output_all_steps=True)
    assert o.output.shape == (2, 4, 1, 2, 2) and tuple(o.hidden_states[0].shape) == (2, 2, 3)
    o = rollout(M(), x, 2)                 # without output_all_steps: the last step's hidden states, the mean loss
    assert float(o.loss) == 1.5 and tuple(o.hidden_states[0].shape) == (2, 3)
