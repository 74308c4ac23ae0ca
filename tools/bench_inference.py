"""Forward-only (inference, no labels) samples/s of Poseidon-B at batch 64 — what scOT/inference.py's rollouts cost per step.
usage: python tools/bench_inference.py [batch]"""
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd.config import preset
from scOT.model import ScOT

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = preset("B", image_size=128, num_channels=4, num_out_channels=4, channel_slice_list_normalized_loss=[0, 1, 3, 4])
torch.manual_seed(1234)
m = ScOT(cfg, compute="fp16").to("cuda").eval()
pv, t = torch.randn(B, 4, 128, 128, device="cuda"), torch.rand(B, device="cuda")
with torch.no_grad():
    for _ in range(4):
        out = m(pixel_values=pv, time=t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        out = m(pixel_values=pv, time=t)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
print(f"inference Poseidon-B batch {B}: {ms:.2f} ms/forward = {B / ms * 1e3:.0f} samples/s")
