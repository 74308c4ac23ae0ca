"""Where does the slowdown of a model created AFTER a larger one sit: in the host's enqueue of the recorded step or on the GPU?"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd.config import preset
from poseidon_amd import ops
from scOT.model import ScOT

cfg = preset("B", image_size=128, num_channels=4, num_out_channels=4, channel_slice_list_normalized_loss=[0, 1, 3, 4])
seq = [int(x) for x in (sys.argv[1:] or ["16", "64", "16"])]
for i, B in enumerate(seq):
    torch.manual_seed(0)
    model = ScOT(cfg, compute="fp16").to("cuda")
    kw = dict(pixel_values=torch.randn(B, 4, 128, 128, device="cuda"), time=torch.rand(B, device="cuda"), labels=torch.randn(B, 4, 128, 128, device="cuda"))
    for _ in range(5):
        model.zero_grad(overlap=True); model(**kw).loss.backward()
    torch.cuda.synchronize()
    eng = model._engine
    ent = list(eng._taped.values())[0]
    prev = ops.use(eng.lib_kind)
    res = []
    for name, seg in (("fwd", (ent["fwd"], ent["fwd_c"])), ("bwd", ent["bwd"][True])):
        hs, gs = [], []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng._replay(*seg)
            t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            hs.append((t1 - t0) * 1e3); gs.append((t2 - t0) * 1e3)
        res.append(f"{name}: host {min(hs):5.2f} ms, until done {min(gs):5.2f} ms, entries {len(seg[0])}")
    ops.use(prev)
    ws = {k: v.numel() >> 20 for k, v in ops._workspace.items()}
    free, total = torch.cuda.mem_get_info()
    print(f"model {i} batch {B}: " + "; ".join(res) + f"; workspaces MB {ws}; reserved {torch.cuda.memory_reserved() >> 20} MB; streams side {eng.side.cuda_stream:#x}", flush=True)
    del model, ent, eng
    torch.cuda.empty_cache()
