"""`output_attentions=True` from the REAL reference (run here only): ScOT.forward returns `decoder_output.attentions +
encoder_outputs.attentions` (reference model.py:1497-1501), where every STAGE contributes the attention probabilities of its LAST
block (`stage_outputs += layer_outputs[1:]`, model.py:859-860, 959-960; collected at model.py:1084-1085, 1225-1226): one
[B·nW, heads, N, N] tensor per stage, decoder stages first.  Tiny config (shifted windows at the first resolution), trained-like
parameters, batch 2; also the positional tuple of `return_dict=False` with attentions on.

usage: python tests/golden/make_attentions_fixture.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf  # noqa: E402,F401  (installs the API-drift shim and imports the reference)
from make_fixtures import TINY, build, save  # noqa: E402
from poseidon_amd.synth import synth_inputs  # noqa: E402


def main():
    torch.manual_seed(0)
    kw = dict(TINY)
    cfg, model = build(kw, "trained")
    model.eval()
    pv, t, lab = synth_inputs(2, kw["num_channels"], kw["num_out_channels"], kw["image_size"], "smooth")
    with torch.no_grad():
        out = model(pixel_values=pv, time=t, labels=lab, output_attentions=True)
        tup = model(pixel_values=pv, time=t, labels=lab, output_attentions=True, output_hidden_states=True, return_dict=False)
    res = {"loss": out.loss.numpy(), "output": out.output.numpy()}
    for i, a in enumerate(out.attentions):
        res[f"attn:{i}"] = a.numpy()
        print(i, tuple(a.shape), float(a.sum(-1).mean()))
    # layout of the positional tuple: which positions hold tensors / tuples and how long the tuples are
    res["tuple_layout"] = torch.tensor([(-1 if torch.is_tensor(x) else len(x)) for x in tup]).numpy()
    save("tiny_attentions", res, dict(cfg=kw, regime="trained", batch=2, kind="smooth", n_attn=len(out.attentions)))


if __name__ == "__main__":
    main()
