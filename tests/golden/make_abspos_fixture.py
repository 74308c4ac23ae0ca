"""`use_absolute_embeddings=True` from the REAL reference (run here only): ScOTEmbeddings adds a learned (1, L, C) position table
after the embedding norm (reference model.py:333-339, 361-362).  Tiny config, trained-like parameters (the table itself is a
closed-form parameter like every other tensor), batch 2; output, loss and all gradients.

usage: python tests/golden/make_abspos_fixture.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf  # noqa: E402,F401  (installs the API-drift shim and imports the reference)
from make_fixtures import TINY, build, run, save  # noqa: E402


def main():
    torch.manual_seed(0)
    kw = dict(TINY, use_absolute_embeddings=True)
    cfg, model = build(kw, "trained")
    assert float(model.embeddings.position_embeddings.abs().sum()) > 0
    res, _ = run(model, kw, batch=2)
    save("tiny_abspos", res, dict(cfg=kw, regime="trained", batch=2, kind="smooth"))


if __name__ == "__main__":
    main()
