"""Pins for the dataset RECIPES from the reference's own readers (build container only).

The reference readers import h5py, which this image does not have; they use it only as `h5py.File(path, "r")[key][index]`.  This
script registers tests/golden/synth_h5.py's SynthFile under the module name `h5py` (a data source, not reference code), imports
`/root/reference/scOT/problems` unchanged, builds every dataset its registry names through the reference's `get_dataset`, runs the
reference's `__getitem__` on a few samples per split and records summaries (tests/golden/synth_h5.py:summary) — the channel order,
normalisation constants, constant planes, masks, time normalisation and split offsets all enter those numbers.

usage: python tests/golden/make_dataset_recipe_pins.py   ->  tests/golden/dataset_recipe_pins.json
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth_h5  # noqa: E402

fake = types.ModuleType("h5py")
fake.File = synth_h5.SynthFile
sys.modules["h5py"] = fake
sys.path.insert(0, "/root/reference")
from scOT.problems.base import get_dataset  # noqa: E402  (the reference)

NAMES = (
    [f"fluids.incompressible.{n}{t}" for n in ("BrownianBridge", "Gaussians", "ShearLayer", "Sines", "PiecewiseConstants", "VortexSheet")
     for t in ("", ".tracer")]
    + ["fluids.incompressible.Sines.out", "fluids.incompressible.forcing.KolmogorovFlow"]
    + [f"fluids.compressible.{n}{t}" for n in ("RiemannKelvinHelmholtz", "RiemannCurved", "Riemann", "KelvinHelmholtz", "Gaussians")
       for t in ("", ".tracer")]
    + ["fluids.compressible.Riemann.out", "fluids.compressible.RichtmyerMeshkov", "fluids.compressible.gravity.RayleighTaylor",
       "fluids.compressible.gravity.RayleighTaylor.tracer", "fluids.compressible.gravity.RayleighTaylor.out",
       "fluids.compressible.steady.Airfoil", "fluids.compressible.steady.Airfoil.time",
       "elliptic.poisson.Gaussians", "elliptic.poisson.Gaussians.time", "elliptic.Helmholtz", "elliptic.Helmholtz.time",
       "wave.Layer", "wave.Layer.out", "wave.Gaussians", "reaction_diffusion.AllenCahn", "reaction_diffusion.AllenCahn.out"]
)
EXTRA = [   # (name, keyword arguments) beyond the defaults
    ("fluids.incompressible.Sines", dict(just_velocities=True)),
    ("fluids.incompressible.Gaussians.tracer", dict(just_velocities=True)),
    ("fluids.incompressible.forcing.KolmogorovFlow", dict(just_velocities=True)),
    ("fluids.incompressible.BrownianBridge", dict(resolution=64)),
    ("fluids.compressible.Riemann", dict(max_num_time_steps=4, time_step_size=3, fix_input_to_time_step=2)),
    ("wave.Layer", dict(max_num_time_steps=5, time_step_size=2, allowed_time_transitions=[1, 3])),
]


def record(sample):
    out = {}
    for k, v in sample.items():
        if k in ("pixel_values", "labels"):
            out[k] = synth_h5.summary(v.numpy())
        elif k == "pixel_mask":
            m = v.numpy()
            out[k] = {"shape": list(m.shape), "dtype": str(v.dtype), "count": int(m.sum()),
                      "wsum": synth_h5.summary(m.astype(np.float32))["wsum"] if m.ndim == 3 else None,
                      "values": m.astype(int).tolist() if m.ndim == 1 else None}
        else:
            out[k] = float(v)
    return out


def main():
    pins = []
    for name, kw in [(n, {}) for n in NAMES] + EXTRA:
        for which, ntraj in (("train", 4), ("val", -1), ("test", 3)):
            try:
                ds = get_dataset(name, which=which, num_trajectories=ntraj, data_path="/synthetic", **kw)
            except (ValueError, AssertionError, TypeError, NotImplementedError) as e:     # combinations the reference refuses
                pins.append(dict(name=name, kw=kw, which=which, num_trajectories=ntraj, raises=type(e).__name__))
                continue
            n = len(ds)
            idx = sorted({0, min(n - 1, 9), n - 1}) if which == "train" else [n // 2]
            inner = getattr(ds, "dataset", ds)
            pins.append(dict(name=name, kw=kw, which=which, num_trajectories=ntraj, length=n, input_dim=int(ds.input_dim),
                             output_dim=int(ds.output_dim), channel_slice_list=list(ds.channel_slice_list), start=int(inner.start),
                             idx=idx, samples=[record(ds[i]) for i in idx]))
    with open(os.path.join(HERE, "dataset_recipe_pins.json"), "w") as f:
        json.dump(pins, f)
    print(len(pins), "pins,", os.path.getsize(os.path.join(HERE, "dataset_recipe_pins.json")) // 1024, "KB")


if __name__ == "__main__":
    torch.manual_seed(0)
    main()
