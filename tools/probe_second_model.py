"""Is the SECOND model of a process slower than the first?  (bench.py's other_configs: B@256 46 ms against 30 ms alone.)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd.config import preset
from poseidon_amd import streams
from scOT.model import ScOT

cfg = preset("B", image_size=128, num_channels=4, num_out_channels=4, channel_slice_list_normalized_loss=[0, 1, 3, 4])
for i, B in enumerate([16, 16, 64, 16]):
    torch.manual_seed(0)
    model = ScOT(cfg, compute="fp16").to("cuda")
    kw = dict(pixel_values=torch.randn(B, 4, 128, 128, device="cuda"), time=torch.rand(B, device="cuda"), labels=torch.randn(B, 4, 128, 128, device="cuda"))
    def run(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            model.zero_grad(overlap=True); model(**kw).loss.backward()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
    run(4)
    e, w = run(10)
    eng = model._engine
    st = [(v.get("state"), sorted(v.get("bwd", {}).keys())) for v in eng._taped.values()]
    ov = streams.overlaps(torch.cuda.current_stream(), eng.side) and streams.overlaps(eng.side, torch.cuda.current_stream())
    print(f"model {i} batch {B}: enqueue {e:6.2f} wall {w:6.2f} ms/step; tapes {st}; side stream {eng.side.cuda_stream:#x} overlaps main: {ov}", flush=True)
    if os.environ.get("KEEP") != "1":
        del model
        torch.cuda.empty_cache()
