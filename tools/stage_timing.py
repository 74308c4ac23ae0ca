"""Per-stage GPU time of one eager fwd+bwd step (HIP events at the stage boundaries of ScOTEngine).

usage: SCOT_STAGE_TIMING=1 [SCOT_SIDE_STREAM=0] python tools/stage_timing.py [--model B --batch 64]
"""
import argparse
import os
import sys

os.environ["SCOT_STAGE_TIMING"] = "1"
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd.config import preset  # noqa: E402
from scOT.model import ScOT  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="B")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--compute", default="fp16")
    a = ap.parse_args()
    cfg = preset(a.model, image_size=a.size, num_channels=4, num_out_channels=4, channel_slice_list_normalized_loss=[0, 1, 3, 4])
    torch.manual_seed(1234)
    model = ScOT(cfg, compute=a.compute).to("cuda")
    B = a.batch
    inp = dict(pixel_values=torch.randn(B, 4, a.size, a.size, device="cuda"), labels=torch.randn(B, 4, a.size, a.size, device="cuda"),
               time=torch.randint(0, 8, (B,), device="cuda").float() / 10.0)
    model(**inp).loss.backward()   # builds the engine
    eng = model._engine
    acc = {}
    order = []
    for it in range(6):
        eng.marks.clear()
        model.zero_grad()
        out = model(pixel_values=inp["pixel_values"], time=inp["time"], labels=inp["labels"])
        out.loss.backward()
        torch.cuda.synchronize()
        if it < 2:
            continue
        m = eng.marks
        for (la, ea), (lb, eb) in zip(m[:-1], m[1:]):
            if la not in acc:
                acc[la] = 0.0
                order.append(la)
            acc[la] += ea.elapsed_time(eb) / 4
    tot = sum(acc.values())
    for la in order:
        print(f"{la:14s} {acc[la]:8.3f} ms  {100 * acc[la] / tot:5.1f} %")
    print(f"{'total':14s} {tot:8.3f} ms")


if __name__ == "__main__":
    main()
