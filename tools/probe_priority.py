"""Does a HIGH-priority main stream protect the backward's dependent chain from the weight-gradient stream?  Poseidon-B batch 64 step on the default
stream (priority 0 = torch's lowest) against the same step issued on a priority -1 stream (the side stream stays at 0); rounds 1 / 4 tried stream
priorities before the queue-sharing bug of streams.py was found, hence the re-check."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd.config import preset
from scOT.model import ScOT

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = preset("B", image_size=128, num_channels=4, num_out_channels=4, channel_slice_list_normalized_loss=[0, 1, 3, 4])
for prio in (0, -1, 0, -1):
    torch.manual_seed(0)
    model = ScOT(cfg, compute="fp16").to("cuda")
    kw = dict(pixel_values=torch.randn(B, 4, 128, 128, device="cuda"), time=torch.rand(B, device="cuda"), labels=torch.randn(B, 4, 128, 128, device="cuda"))
    main = torch.cuda.Stream(priority=-1) if prio == -1 else torch.cuda.current_stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(main):
        def loop(n):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n):
                model.zero_grad(overlap=True); model(**kw).loss.backward()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        loop(5)
        print(f"main stream priority {prio:2d}: {loop(20):.3f} ms/step (side stream {model._engine.side.cuda_stream:#x}, main {main.cuda_stream:#x})", flush=True)
    del model
    torch.cuda.empty_cache()
