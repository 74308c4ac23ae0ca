"""Optimizer step of Poseidon-B (157.7 M parameters in 1580 tensors): FusedAdamW (arena, 3 launches) vs torch.optim.AdamW
on the reference's parameter groups + clip_grad_norm_.   python tools/bench_optimizer.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd.config import preset  # noqa: E402
from scOT.model import ScOT  # noqa: E402
from scOT.trainer import FusedAdamW, create_optimizer  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def main():
    cfg = preset("B", image_size=128, num_channels=4, num_out_channels=4, channel_slice_list_normalized_loss=[0, 1, 3, 4])
    model = ScOT(cfg, compute="bf16").to("cuda")
    B = 8
    kw = dict(pixel_values=torch.randn(B, 4, 128, 128, device="cuda"), labels=torch.randn(B, 4, 128, 128, device="cuda"),
              time=torch.rand(B, device="cuda"))
    model(**kw).loss.backward()
    gk = dict(learning_rate_embedding_recovery=1e-4, learning_rate_time_embedding=2e-4)
    fused = FusedAdamW(model, lr=5e-4, weight_decay=1e-6, max_grad_norm=5.0, **gk)
    stock = create_optimizer(model, 5e-4, weight_decay=1e-6, **gk)

    def stock_step():
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        stock.step()
    print(f"FusedAdamW (clip + step)            : {timed(fused.step):7.2f} ms")
    print(f"torch AdamW + clip_grad_norm_ (stock): {timed(stock_step):7.2f} ms")


if __name__ == "__main__":
    main()
