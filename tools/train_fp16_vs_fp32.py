"""Does training in the timed mode (fp16 operands, fp32 accumulation and master weights, dynamic gradient scale) follow the reference's fp32
recipe (scOT/train.py:277-323: AdamW, clip 5.0, lr 5e-5, fp16=False) on the HEADLINE model?  Poseidon-B, 128 x 128 x 4, batch 16, trained-like
parameters, STEPS fused-AdamW steps over a cycle of NB different synthetic batches, once per mode from the same state.
Prints the two loss trajectories, their largest relative gap, the fp16 run's overflow / skipped-step counts and the relative distance of the
final weights.   usage: python tools/train_fp16_vs_fp32.py [steps=60] [batches=6] [model=B]"""
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd.config import preset
from poseidon_amd.geometry import param_shapes
from poseidon_amd.synth import synth_inputs, synth_state_dict
from scOT.model import ScOT
from scOT.trainer import FusedAdamW

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 6
TAG = sys.argv[3] if len(sys.argv) > 3 else "B"
BATCH = 16
cfg = preset(TAG, image_size=128, num_channels=4, num_out_channels=4, channel_slice_list_normalized_loss=[0, 1, 3, 4])
sd = synth_state_dict(param_shapes(cfg), "trained")
pv, t, lab = synth_inputs(BATCH * NB, 4, 4, 128, "smooth")
batches = [dict(pixel_values=pv[i * BATCH:(i + 1) * BATCH].cuda(), time=t[i * BATCH:(i + 1) * BATCH].cuda(),
                labels=lab[i * BATCH:(i + 1) * BATCH].cuda()) for i in range(NB)]
traj, final, info = {}, {}, {}
for compute in ("fp32", "fp16"):
    model = ScOT(cfg, compute=compute)
    model.load_state_dict(sd)
    model = model.cuda()
    opt = FusedAdamW(model, lr=5e-5, weight_decay=0.01, max_grad_norm=5.0)
    losses = []
    for s in range(STEPS):
        opt.zero_grad(overlap=True)
        out = model(**batches[s % NB])
        out.loss.backward()
        opt.step()
        losses.append(float(out.loss.detach()))
    torch.cuda.synchronize()
    traj[compute], final[compute] = np.array(losses), model.flat_parameters().clone()
    info[compute] = dict(grad_overflow=int(model._engine.grad_overflow or 0), skipped_steps=int(opt.skipped_steps() or 0))
    del model, opt
    torch.cuda.empty_cache()
a, b = traj["fp32"], traj["fp16"]
gap = np.abs(a - b) / a
print(f"Poseidon-{TAG} batch {BATCH}, {STEPS} AdamW steps over {NB} batches (lr 5e-5, clip 5.0, weight decay 0.01)")
print("step   fp32 loss   fp16 loss   rel gap")
for s in sorted(set(list(range(0, STEPS, max(1, STEPS // 12))) + [STEPS - 1])):
    print(f"{s:4d}  {a[s]:10.5f}  {b[s]:10.5f}  {gap[s]:.2e}")
epoch = lambda x, k: float(np.mean(x[k * NB:(k + 1) * NB]))
print(f"mean loss of the first / last cycle of {NB} batches: fp32 {epoch(a, 0):.4f} -> {epoch(a, STEPS // NB - 1):.4f}; fp16 {epoch(b, 0):.4f} -> {epoch(b, STEPS // NB - 1):.4f}")
print(f"max rel gap {gap.max():.2e} (median {np.median(gap):.2e}); final weights rel-L2 fp16 vs fp32 {float((final['fp16'] - final['fp32']).norm() / final['fp32'].norm()):.2e}; fp16 run: {info['fp16']}")
