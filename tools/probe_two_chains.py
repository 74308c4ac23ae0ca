"""What-if: does the chip fill up when TWO half-batch steps run concurrently (two engines, two main streams + their side streams)?
Poseidon-B, 2 x batch 32 against 1 x batch 64 (fwd + bwd, fp16 build).  usage: python tools/probe_two_chains.py [total_batch] [fwd]
("fwd": forward only under no_grad — the phase in which the weight-gradient stream is idle)"""
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd.config import preset
from scOT.model import ScOT

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
FWD_ONLY = len(sys.argv) > 2 and sys.argv[2] == "fwd"
cfg = preset("B", image_size=128, num_channels=4, num_out_channels=4, channel_slice_list_normalized_loss=[0, 1, 3, 4])


def make(batch, seed):
    torch.manual_seed(1234)
    m = ScOT(cfg, compute="fp16").to("cuda")
    if FWD_ONLY:
        m.eval()
    torch.manual_seed(seed)
    kw = dict(pixel_values=torch.randn(batch, 4, 128, 128, device="cuda"), labels=torch.randn(batch, 4, 128, 128, device="cuda"),
              time=torch.rand(batch, device="cuda"))
    if FWD_ONLY:
        del kw["labels"]
    return m, kw


def step(m, kw):
    if FWD_ONLY:
        with torch.no_grad():
            m(**kw)
        return
    m.zero_grad(overlap=True)
    out = m(**kw)
    out.loss.backward()


def timeit(fn, n=10):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    t_enq = time.perf_counter() - t
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, t_enq / n * 1e3


full = make(B, 1)
ms, enq = timeit(lambda: step(*full))
print(f"1 x batch {B}: {ms:.2f} ms/step (host enqueue {enq:.2f})")
del full
torch.cuda.empty_cache()
for nch in (2, 3):
    chains = [make(B // nch, 10 + i) for i in range(nch)]
    streams = [torch.cuda.Stream() for _ in range(nch)]

    def both():
        for (m, kw), s in zip(chains, streams):
            with torch.cuda.stream(s):
                step(m, kw)
    ms2, enq2 = timeit(both)
    one, _ = timeit(lambda: step(*chains[0]))
    print(f"{nch} x batch {B // nch} on {nch} streams: {ms2:.2f} ms per {B // nch * nch} samples (host enqueue {enq2:.2f}); one batch-{B // nch} chain alone: {one:.2f} ms")
    del chains
    torch.cuda.empty_cache()
