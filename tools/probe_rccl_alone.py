"""Why does a fresh process that creates an RCCL communicator AFTER its first steps run the overlapped exchange at half speed?
(`test_overlapped_gradient_exchange_under_a_one_rank_rccl_group` alone: 22.6 vs 11.2 ms; inside the suite and in bench.py: within 2 %.)
Phases, Poseidon-B batch 16 fp16, one process: bare -> after init_process_group -> reducer attached (torch / native backend), with the
fork / join measurement of every stream pair at each phase.  usage: python tools/probe_rccl_alone.py [early|late|refresh] [torch|native]
("refresh": as late, but the engine's weight-gradient stream is dropped and re-picked after the communicator exists)"""
import os
import socket
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from poseidon_amd import streams
from poseidon_amd.config import preset
from poseidon_amd.dp import OverlappedGradAllReducer
from scOT.model import ScOT

WHEN = sys.argv[1] if len(sys.argv) > 1 else "late"
BACKEND = sys.argv[2] if len(sys.argv) > 2 else "torch"
cfg = preset("B", image_size=128, num_channels=4, num_out_channels=4, channel_slice_list_normalized_loss=[0, 1, 3, 4])


def pg():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    x = torch.ones(1024, device="cuda")
    dist.all_reduce(x)
    torch.cuda.synchronize()


if WHEN == "early" and BACKEND == "torch":
    pg()
torch.manual_seed(1234)
model = ScOT(cfg, compute="fp16").to("cuda")
kw = dict(pixel_values=torch.randn(16, 4, 128, 128, device="cuda"), labels=torch.randn(16, 4, 128, 128, device="cuda"), time=torch.rand(16, device="cuda"))


def timed(n=6):
    for _ in range(3):
        model.zero_grad()
        model(**kw).loss.backward()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        model.zero_grad()
        model(**kw).loss.backward()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def pairs(tag, extra=()):
    main, side = torch.cuda.current_stream(), model._engine.side_stream()
    named = [("main", main), ("side", side)] + list(extra)
    out = []
    for i in range(len(named)):
        for j in range(i + 1, len(named)):
            out.append(f"{named[i][0]}|{named[j][0]}={'ok' if streams.overlaps(named[i][1], named[j][1]) else 'SERIAL'}")
    print(f"  [{tag}] " + " ".join(out), flush=True)


print(f"bare: {timed():.2f} ms", flush=True)
pairs("bare")
if WHEN in ("late", "refresh") and BACKEND == "torch":
    pg()
    print(f"after init_process_group: {timed():.2f} ms", flush=True)
    pairs("after init")
if BACKEND == "native":
    from poseidon_amd.dp import native_init
    native_init(None)
    print(f"after scot_dp_init: {timed():.2f} ms", flush=True)
if WHEN == "refresh":
    streams._cache.clear()
    model._engine.side = None
    model._engine.reset_tapes()
    print(f"fresh weight-gradient stream: {timed():.2f} ms", flush=True)
for wire in ("fp32", "bf16"):
    red = OverlappedGradAllReducer(model, dist if BACKEND == "torch" else None, wire=wire, backend=BACKEND)
    red.attach()
    print(f"{BACKEND} backend, {wire} wire attached: {timed():.2f} ms", flush=True)
    pairs("attached", [("comm", red.comm_stream)])
    red.timing = []
    ms = timed()
    print(f"   again {ms:.2f} ms, comm-stream time per step {red.comm_ms() / 9:.2f} ms", flush=True)
    red.detach()
print(f"detached: {timed():.2f} ms", flush=True)
if dist.is_initialized():
    dist.destroy_process_group()
