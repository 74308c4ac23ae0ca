"""Experiment (not product code): does the chip have room for a second independent chain?  One model at batch 64 vs two
independent models at batch 32 stepping concurrently from two Python threads on their own streams (same total samples)."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd import ops  # noqa: E402
from poseidon_amd.config import preset  # noqa: E402
from scOT.model import ScOT  # noqa: E402

_cur = [0]
_orig_ws = ops.workspace
_ws = {}


def workspace(need=0):
    key = (_cur[0], ops._slot)
    w = _ws.get(key)
    if w is None:
        w = _ws[key] = torch.empty((96 << 20), dtype=torch.uint8, device="cuda")
    return w


ops.workspace = workspace


def make(batch, idx, stream):
    cfg = preset("B", image_size=128, num_channels=4, num_out_channels=4, channel_slice_list_normalized_loss=[0, 1, 3, 4])
    torch.manual_seed(1234)
    m = ScOT(cfg).to("cuda")
    kw = dict(pixel_values=torch.randn(batch, 4, 128, 128, device="cuda"), labels=torch.randn(batch, 4, 128, 128, device="cuda"),
              time=torch.rand(batch, device="cuda"))

    def step():
        m.zero_grad()
        m(**kw).loss.backward()
    _cur[0] = idx
    with torch.cuda.stream(stream):
        for _ in range(4):
            step()
    torch.cuda.synchronize()
    return step


def run(steps_fns, streams, n=10):
    def loop(fn, st):
        with torch.cuda.stream(st):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=loop, args=(f, s)) for f, s in zip(steps_fns, streams)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    one = make(64, 0, s0)
    print(f"1 x batch 64: {run([one], [s0]):.2f} ms/step")
    del one
    torch.cuda.empty_cache()
    a, b = make(32, 1, s0), make(32, 2, s1)
    print(f"1 x batch 32 alone: {run([a], [s0]):.2f} ms/step")
    print(f"2 x batch 32 concurrently: {run([a, b], [s0, s1]):.2f} ms per 64 samples")


if __name__ == "__main__":
    main()
