"""Data-parallel gradient exchange logic on CPU: world_size 2, gloo backend (the GPU path uses the same code with the
`nccl` (= RCCL) backend).  Covers: mean all-reduce of the flat gradient arena in chunks, fp32 and bf16 wire formats,
parameter broadcast, and that the backward-order ranges are contiguous, disjoint and cover every parameter."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from poseidon_amd.arena import Arena
from poseidon_amd.config import ScOTConfig, preset
from poseidon_amd.dp import GradAllReducer, backward_order_groups, group_ranges
from poseidon_amd.geometry import param_shapes

TINY = dict(image_size=32, patch_size=4, num_channels=4, num_out_channels=4, embed_dim=16, depths=[2, 2], num_heads=[1, 2],
            skip_connections=[1, 0], window_size=4, use_conditioning=True)


class _FakeModel:
    def __init__(self, cfg):
        self.config = cfg
        self._arena = Arena(param_shapes(cfg), "cpu")

    def flat_grads(self):
        return self._arena.grad

    def flat_parameters(self):
        return self._arena.data

    dirty = 0

    def mark_weights_dirty(self):       # (ScOT's: a collective writes the arena without moving torch's version counter)
        self.dirty += 1


def test_backward_order_ranges_cover_arena():
    for cfg in (ScOTConfig(**TINY), preset("B", image_size=128, num_channels=4, num_out_channels=4)):
        ar = Arena(param_shapes(cfg), "cpu")
        rngs = group_ranges(ar, backward_order_groups(cfg))
        assert [r[0] for r in rngs][0] == "patch_recovery." and rngs[-1][0] == "embeddings."
        if cfg.embed_dim == 96:     # Poseidon-B: the two C = 768 stages are announced in halves of four blocks (dp.stage_split)
            keys = [r[0] for r in rngs]
            assert keys[4:8] == ["decoder.layers.0.blocks[4:]", "decoder.layers.0.blocks[:4]", "encoder.layers.3.blocks[4:]", "encoder.layers.3.blocks[:4]"]
            assert max(e - s for _, s, e in rngs) * 4 < 125e6      # no range above 119 MB (was 233 MB)
        spans = sorted((s, e) for _, s, e in rngs)
        for (s0, e0), (s1, e1) in zip(spans, spans[1:]):
            assert e0 <= s1  # disjoint
        covered = torch.zeros(ar.size, dtype=torch.bool)
        for _, s, e in rngs:
            covered[s:e] = True
        for n in ar.shapes:
            o = ar.offsets[n]
            assert bool(covered[o:o + ar.numel(n)].all()), n
        # fused qkv views are inside their stage's range
        assert "encoder.layers.0.blocks.0.attention.self.qkv_weight" in ar.offsets


def _emu_ops():
    """The wire pack / unpack are HIP kernels (scot_dp_pack / scot_dp_unpack): on CPU tensors they run from the emulated build."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "hipemu"))
    import emu_session
    from poseidon_amd import ops
    lib = emu_session.load_emu()
    ops.L, ops.stream = (lambda: lib), (lambda: None)
    ops.ptr = lambda t: None if t is None else t.data_ptr()


def _free_port():
    """a port the OS hands out for 127.0.0.1 (a fixed one shared by consecutive tests can still sit in TIME_WAIT)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _worker(rank, world, port, wire, collective, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if wire == "bf16":
        _emu_ops()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = ScOTConfig(**TINY)
    m = _FakeModel(cfg)
    n = m._arena.size
    base = torch.arange(n, dtype=torch.float32) / n
    m._arena.grad.copy_(base * (rank + 1))           # rank r holds (r+1)*base → mean = 1.5*base
    m._arena.data.fill_(float(rank + 7))
    red = GradAllReducer(m, dist, wire=wire, chunk_mb=1, collective=collective)
    red.chunk = 1000                                  # force many chunks (1000 % (2 * 8) != 0: rs_ag pads every chunk)
    v0 = m._arena.data._version
    red.broadcast_parameters(src=0)
    assert float(m._arena.data.min()) == 7.0 and float(m._arena.data.max()) == 7.0
    assert m.dirty == 1          # ... so the engine re-casts its 16-bit copies whatever torch's counter did ({m._arena.data._version - v0})
    red.allreduce()
    tol = 1e-6 if wire == "fp32" else 8e-3
    assert torch.allclose(m._arena.grad, 1.5 * base, rtol=tol, atol=tol * 0.01), (rank, wire)
    # range-wise reduction in backward order gives the same result
    m._arena.grad.copy_(base * (rank + 1))
    for _, s, e in red.ranges_in_backward_order():
        red.reduce_range(s, e)
    covered = torch.zeros(n, dtype=torch.bool)
    for _, s, e in red.ranges_in_backward_order():
        covered[s:e] = True
    assert torch.allclose(m._arena.grad[covered], (1.5 * base)[covered], rtol=tol, atol=tol * 0.01)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        open(out, "w").write("ok")


@pytest.mark.parametrize("wire,collective", [("fp32", "allreduce"), ("bf16", "allreduce"), ("fp32", "rs_ag"), ("bf16", "rs_ag")])
def test_two_rank_mean_allreduce(tmp_path, wire, collective):
    if wire == "bf16":
        import shutil
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))
        import build_emu
        if not (os.path.exists(build_emu.CLANG) or shutil.which(build_emu.CLANG)):
            pytest.skip("no host clang with __bf16 vector support")
        build_emu.build_cached()                              # once, before the two ranks race for it
    port = _free_port()
    out = str(tmp_path / "ok")
    mp.spawn(_worker, args=(2, port, wire, collective, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


# ------------------------------------------------------------------------------------------------------------------------
# The real engine under data parallelism, on the CPU emulation of the kernels (tests/hipemu): each rank runs forward + backward
# of its shard and the engine's `on_grads_final(prefix)` callback — the hook the overlapped RCCL exchange hangs on — reduces that
# prefix's arena range IMMEDIATELY.  If a range were announced before the backward had finished writing it, the contributions
# accumulated afterwards would stay un-averaged and the result would differ from the mean of the two ranks' gradients.
def _engine_worker(rank, world, port, out, depths=(2, 2)):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "hipemu"))
    sys.path.insert(0, here)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SCOT_SIDE_STREAM="0", SCOT_TAPE="0")
    import emu_session
    from poseidon_amd import ops
    from poseidon_amd.synth import synth_inputs, synth_state_dict
    from scOT.model import ScOT
    lib = emu_session.load_emu()
    ws = torch.empty(32 << 20, dtype=torch.uint8)
    ops.L, ops.stream, ops.workspace = (lambda: lib), (lambda: None), (lambda need=0: ws)
    ops.ptr = lambda t: None if t is None else t.data_ptr()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = ScOTConfig(**dict(TINY, depths=list(depths), mlp_ratio=4.0, qkv_bias=True, p=1, channel_slice_list_normalized_loss=[0, 1, 3, 4],
                            drop_path_rate=0.0))      # (stochastic depth would make the two steps below differ)
    model = ScOT(cfg, compute="fp32")
    model.load_state_dict(synth_state_dict(param_shapes(cfg), "trained"))
    model._ensure_arena(torch.device("cpu"))
    pv, t, lab = synth_inputs(4, 4, 4, 32, "smooth")
    sl = slice(2 * rank, 2 * rank + 2)                       # this rank's shard of the global batch of 4
    eng = model._engine

    def step():
        model._arena.grad.zero_()
        loss, _, tape = eng.forward(pv[sl].contiguous(), t[sl].contiguous(), lab[sl].contiguous(), None, train=True)
        model._prepare_grads()
        eng.backward(tape, torch.ones(1), None)
        return model._arena.grad.clone()

    local = step()                                           # no exchange
    both = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    expect = sum(both) / world
    assert not torch.equal(both[0], both[1])                 # the shards really differ
    red = GradAllReducer(model, dist, wire="fp32", chunk_mb=1)
    ranges = {p: (s, e) for p, s, e in red.ranges_in_backward_order()}
    seen = []

    def on_final(prefix):
        seen.append(prefix)
        s, e = ranges[prefix]
        red.reduce_range(s, e)

    eng.on_grads_final = on_final
    got = step()
    # (no side stream on the CPU emulation: the skip blocks' backward runs in line, before the encoder stages)
    assert seen == [p for p in backward_order_groups(cfg, skips_on_side=False) if p in ranges], seen
    if tuple(depths) == (2, 4):      # the widest stages are announced in halves: the upper blocks (+ the resampling layer) first
        assert seen[2:4] == ["decoder.layers.0.blocks[2:]", "decoder.layers.0.blocks[:2]"] and "encoder.layers.1.blocks[2:]" in seen, seen
    err = float((got - expect).norm() / expect.norm())
    bad = [n for n in model._arena.shapes
           if not torch.allclose(model._arena.gview(n), expect[model._arena.offsets[n]:model._arena.offsets[n] + model._arena.numel(n)]
                                 .view(model._arena.gview(n).shape), rtol=1e-5, atol=1e-7)]
    assert err < 1e-6 and not bad, (rank, err, bad[:12], len(bad))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        open(out, "w").write(f"ok {err:.2e}")


def test_engine_ranges_are_final_when_announced(tmp_path):
    import shutil
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))
    import build_emu
    if not (os.path.exists(build_emu.CLANG) or shutil.which(build_emu.CLANG)):
        pytest.skip("no host clang with __bf16 vector support")
    build_emu.build_cached()                                  # once, before the two ranks race for it
    out = str(tmp_path / "ok")
    mp.spawn(_engine_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read().startswith("ok")


def test_engine_stage_halves_are_final_when_announced(tmp_path):
    """a model whose widest stages hold > 1/8 of the parameters each (as Poseidon-T / B / L's do): dp.stage_split announces them in two
    halves, the upper one in the middle of the stage's backward — still only after its last gradient writer"""
    import shutil
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))
    import build_emu
    if not (os.path.exists(build_emu.CLANG) or shutil.which(build_emu.CLANG)):
        pytest.skip("no host clang with __bf16 vector support")
    build_emu.build_cached()
    out = str(tmp_path / "ok")
    mp.spawn(_engine_worker, args=(2, _free_port(), out, (2, 4)), nprocs=2, join=True)
    assert open(out).read().startswith("ok")


# ------------------------------------------------------------------------------------------------------------------------
# The same hook through the STEP TAPE with the bf16 wire (round-2 advisor finding): the exchange's pack / unpack launches go through
# `ops`, so while a step is being recorded they must not be logged next to the callback that issues them — a replayed step would
# otherwise pack / sum / unpack every range twice (gradient = mean / N).  Steps 1 (direct), 2 (recorded), 3-5 (replayed) must all
# equal the fp32 mean of the two ranks' stand-alone gradients to bf16 rounding.
def _taped_bf16_worker(rank, world, port, out):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "hipemu"))
    sys.path.insert(0, here)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SCOT_SIDE_STREAM="0", SCOT_TAPE="1")
    import emu_session
    from poseidon_amd import ops
    from poseidon_amd.synth import synth_inputs, synth_state_dict
    from scOT.model import ScOT
    lib = emu_session.load_emu()
    ws = torch.empty(32 << 20, dtype=torch.uint8)
    ops.L = lambda: ops._Recording(lib, ops._recorder) if ops._recorder is not None else lib
    ops.stream, ops.workspace = (lambda: None), (lambda need=0: ws)
    ops.ptr = lambda t: None if t is None else t.data_ptr()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = ScOTConfig(**dict(TINY, mlp_ratio=4.0, qkv_bias=True, p=1, channel_slice_list_normalized_loss=[0, 1, 3, 4], drop_path_rate=0.0))
    model = ScOT(cfg, compute="fp32")
    model.load_state_dict(synth_state_dict(param_shapes(cfg), "trained"))
    model._ensure_arena(torch.device("cpu"))
    eng = model._engine
    assert eng.tape_mode

    def step(k):
        pv, t, lab = synth_inputs(4, 4, 4, 32, "smooth")
        pv = pv * (1.0 + 0.1 * k)                              # different data every step (a replay must not reuse step 2's numbers)
        sl = slice(2 * rank, 2 * rank + 2)
        model._arena.grad.zero_()
        loss, _, tape = eng.forward(pv[sl].contiguous(), t[sl].contiguous(), lab[sl].contiguous(), None, train=True)
        model._prepare_grads()
        eng.backward(tape, torch.ones(1), None)
        return model._arena.grad.clone()

    expect = []
    for k in range(5):                                         # reference: no hook, fp32 mean over ranks after the step
        g = step(k)
        both = [torch.empty_like(g) for _ in range(world)]
        dist.all_gather(both, g)
        expect.append(sum(both) / world)
    eng.reset_tapes()
    red = GradAllReducer(model, dist, wire="bf16", chunk_mb=1)
    ranges = {p: (s, e) for p, s, e in red.ranges_in_backward_order()}
    calls = []

    def on_final(prefix):
        calls.append(prefix)
        s, e = ranges[prefix]
        red.reduce_range(s, e)

    eng.on_grads_final = on_final
    errs = []
    for k in range(5):
        n0 = len(calls)
        got = step(k)
        assert len(calls) - n0 == len(ranges)                 # every range announced exactly once per step, replayed steps included
        errs.append(float((got - expect[k]).norm() / expect[k].norm()))
    states = [e["state"] for e in eng._taped.values()]
    assert states == ["ready"], states                         # steps 3-5 really were replays
    assert max(errs) < 6e-3, (rank, errs)                      # bf16 wire: 2^-9 per element; the double exchange was off by 2x
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        open(out, "w").write("ok " + " ".join(f"{e:.1e}" for e in errs))


def test_taped_step_with_bf16_wire_exchanges_each_range_once(tmp_path):
    import shutil
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))
    import build_emu
    if not (os.path.exists(build_emu.CLANG) or shutil.which(build_emu.CLANG)):
        pytest.skip("no host clang with __bf16 vector support")
    build_emu.build_cached()
    out = str(tmp_path / "ok")
    mp.spawn(_taped_bf16_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read().startswith("ok")


# ------------------------------------------------------------------------------------------------------------------------
# The training driver (poseidon_amd/train.py) under data parallelism: two ranks, the epoch's permutation sharded
# DistributedSampler-style, the mean gradient exchanged after every step -> both replicas hold the SAME parameters after
# training (DDP's invariant), they differ from what either rank would have learnt alone, and evaluation / prediction return the
# whole dataset in order on every rank.
def _trainer_worker(rank, world, port, out):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "hipemu"))
    sys.path.insert(0, here)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SCOT_SIDE_STREAM="0")
    import emu_session
    import scOT.model as M
    from poseidon_amd import ops
    from poseidon_amd.synth import synth_state_dict
    from scOT.trainer import Trainer, TrainingArguments
    from test_trainer_emu_cpu import Samples
    lib = emu_session.load_emu()
    ws = torch.empty(32 << 20, dtype=torch.uint8)
    ops.L, ops.stream, ops.workspace = (lambda: ops._Recording(lib, ops._recorder) if ops._recorder is not None else lib), \
        (lambda: None), (lambda need=0: ws)
    ops.ptr = lambda t: None if t is None else t.data_ptr()
    M._require_hip = lambda t: None
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = ScOTConfig(**dict(TINY, mlp_ratio=4.0, qkv_bias=True, p=1, channel_slice_list_normalized_loss=[0, 1, 3, 4], drop_path_rate=0.0))
    sd = synth_state_dict(param_shapes(cfg), "trained")

    def fit(use_dist):
        model = M.ScOT(cfg, compute="fp32")
        model.load_state_dict(sd)
        if rank == 1 and use_dist:          # a replica that starts elsewhere: the trainer broadcasts rank 0's weights first
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(0.01)
        tr = Trainer(model, TrainingArguments(per_device_train_batch_size=2, per_device_eval_batch_size=2, num_train_epochs=1, learning_rate=1e-3,
                                              lr_scheduler_type="linear", logging_steps=1, max_grad_norm=5.0, dp_exchange="after"),
                     train_dataset=Samples(7, cfg, 0), eval_dataset=Samples(5, cfg, 1))
        if not use_dist:
            tr.dist, tr.world, tr.rank = None, 1, 0
        return tr, tr.train()

    tr, res = fit(True)
    assert res.global_step == 2                      # 7 samples -> 4 per rank (one wrapped) -> 2 batches of 2
    flat = tr.model.flat_parameters().clone()
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1])             # replicas in lock-step
    losses = [h["loss"] for h in tr.state["log_history"] if "loss" in h]
    gathered = [None, None]
    dist.all_gather_object(gathered, losses)
    assert gathered[0] == gathered[1]                # the logged loss is the mean over the ranks
    ev = tr.evaluate()
    pr = tr.predict(Samples(5, cfg, 1), metric_key_prefix="t")
    assert pr.predictions.shape == (5, 4, 32, 32) and pr.metrics["t_loss"] == pytest.approx(ev["eval_loss"], rel=1e-6)
    solo_tr, _ = fit(False)                          # the same recipe without the exchange, on this process alone
    single = solo_tr.predict(Samples(5, cfg, 1), metric_key_prefix="t")
    assert single.predictions.shape == pr.predictions.shape and np.array_equal(single.label_ids, pr.label_ids)   # gathered back in dataset order
    assert not torch.equal(solo_tr.model.flat_parameters(), flat)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        open(out, "w").write("ok")


def test_trainer_two_ranks_stay_in_lock_step(tmp_path):
    import shutil
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))
    import build_emu
    if not (os.path.exists(build_emu.CLANG) or shutil.which(build_emu.CLANG)):
        pytest.skip("no host clang with __bf16 vector support")
    build_emu.build_cached()
    out = str(tmp_path / "ok")
    mp.spawn(_trainer_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"
