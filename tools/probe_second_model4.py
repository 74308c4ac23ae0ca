"""Six models in a row, same size: which ones run slow in the un-synchronised training loop, and what about them differs?"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd.config import preset
from poseidon_amd import ops, streams
from scOT.model import ScOT

cfg = preset("B", image_size=128, num_channels=4, num_out_channels=4, channel_slice_list_normalized_loss=[0, 1, 3, 4])
B = 16
keep = []
for i in range(7):
    torch.manual_seed(0)
    model = ScOT(cfg, compute="fp16").to("cuda")
    kw = dict(pixel_values=torch.randn(B, 4, 128, 128, device="cuda"), time=torch.rand(B, device="cuda"), labels=torch.randn(B, 4, 128, 128, device="cuda"))
    def loop(n, overlap=True, sync=False):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            model.zero_grad(overlap=overlap); model(**kw).loss.backward()
            if sync: torch.cuda.synchronize()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    loop(5)
    a, b, c = loop(10), loop(10, overlap=False), loop(10, sync=True)
    eng = model._engine
    print(f"model {i}: loop {a:6.2f}  eager-fill {b:6.2f}  synced-each-step {c:6.2f} ms/step; side {eng.side.cuda_stream:#x} main {torch.cuda.current_stream().cuda_stream:#x}", flush=True)
    keep.append(model)       # nothing is freed: no allocator reuse between models
