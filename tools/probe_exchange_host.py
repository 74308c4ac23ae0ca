"""Host cost of the overlapped gradient exchange under a 1-rank RCCL group: CPU time to ENQUEUE a step (no synchronisation inside the loop)
and GPU time per step, bare vs with OverlappedGradAllReducer attached, for a short step (Poseidon-B batch 16, ~11 ms) and the headline one (batch 64).
    python tools/probe_exchange_host.py [batch ...]"""
import os
import socket
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd.config import preset  # noqa: E402
from poseidon_amd.dp import OverlappedGradAllReducer  # noqa: E402
from scOT.model import ScOT  # noqa: E402


def main():
    batches = [int(b) for b in sys.argv[1:]] or [16, 64]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    cfg = preset("B", image_size=128, num_channels=4, num_out_channels=4, channel_slice_list_normalized_loss=[0, 1, 3, 4])
    for B in batches:
        torch.manual_seed(0)
        model = ScOT(cfg, compute="fp16").to("cuda")
        kw = dict(pixel_values=torch.randn(B, 4, 128, 128, device="cuda"), time=torch.rand(B, device="cuda"), labels=torch.randn(B, 4, 128, 128, device="cuda"))

        def run(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                model.zero_grad(overlap=True)
                model(**kw).loss.backward()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
        run(4)
        print(f"batch {B}: bare            enqueue {run(10)[0]:6.2f} ms/step   wall {run(10)[1]:6.2f} ms/step", flush=True)
        for wire, chunk in (("fp32", 64), ("fp32", 1024), ("bf16", 64)):
            red = OverlappedGradAllReducer(model, dist, wire=wire, chunk_mb=chunk)
            red.attach()
            run(4)
            e, w = run(10)
            ncalls = 0
            print(f"batch {B}: {wire} wire chunk {chunk:4d} MB  enqueue {e:6.2f} ms/step   wall {w:6.2f} ms/step", flush=True)
            red.detach()
        del model
        torch.cuda.empty_cache()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
