"""cProfile of the training loop of a model created after a larger one (host side of the slowdown)."""
import cProfile, pstats, io, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd.config import preset
from scOT.model import ScOT

cfg = preset("B", image_size=128, num_channels=4, num_out_channels=4, channel_slice_list_normalized_loss=[0, 1, 3, 4])
for i, B in enumerate([16, 64, 16]):
    torch.manual_seed(0)
    model = ScOT(cfg, compute="fp16").to("cuda")
    kw = dict(pixel_values=torch.randn(B, 4, 128, 128, device="cuda"), time=torch.rand(B, device="cuda"), labels=torch.randn(B, 4, 128, 128, device="cuda"))
    def loop(n):
        for _ in range(n):
            model.zero_grad(overlap=True); model(**kw).loss.backward()
    loop(5); torch.cuda.synchronize()
    t0 = time.perf_counter(); loop(10); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"model {i} batch {B}: enqueue {(t1-t0)*100:.2f} wall {(t2-t0)*100:.2f} ms/step", flush=True)
    if i != 1:
        pr = cProfile.Profile(); pr.enable(); loop(10); pr.disable(); torch.cuda.synchronize()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12)
        print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:3000], flush=True)
    del model
    torch.cuda.empty_cache()
