"""Data-parallel gradient exchange logic on CPU: world_size 2, gloo backend (the GPU path uses the same code with the
`nccl` (= RCCL) backend).  Covers: mean all-reduce of the flat gradient arena in chunks, fp32 and bf16 wire formats,
parameter broadcast, and that the backward-order ranges are contiguous, disjoint and cover every parameter."""
import os

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


def test_backward_order_ranges_cover_arena():
    for cfg in (ScOTConfig(**TINY), preset("B", image_size=128, num_channels=4, num_out_channels=4)):
        ar = Arena(param_shapes(cfg), "cpu")
        rngs = group_ranges(ar, backward_order_groups(cfg))
        assert [r[0] for r in rngs][0] == "patch_recovery." and rngs[-1][0] == "embeddings."
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


def _worker(rank, world, port, wire, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = ScOTConfig(**TINY)
    m = _FakeModel(cfg)
    n = m._arena.size
    base = torch.arange(n, dtype=torch.float32) / n
    m._arena.grad.copy_(base * (rank + 1))           # rank r holds (r+1)*base → mean = 1.5*base
    m._arena.data.fill_(float(rank + 7))
    red = GradAllReducer(m, dist, wire=wire, chunk_mb=1)
    red.chunk = 1000                                  # force many chunks
    red.broadcast_parameters(src=0)
    assert float(m._arena.data.min()) == 7.0 and float(m._arena.data.max()) == 7.0
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


@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_two_rank_mean_allreduce(tmp_path, wire):
    port = 29500 + (os.getpid() % 2000) + (0 if wire == "fp32" else 1)
    out = str(tmp_path / "ok")
    mp.spawn(_worker, args=(2, port, wire, out), nprocs=2, join=True)
    assert open(out).read() == "ok"
