"""Dataset front-end (poseidon_amd/data.py): index machinery against pins produced by the reference's own base classes
(tests/golden/make_dataset_pins.py), `get_dataset` names / defaults, the per-dataset sample recipes against a numpy evaluation of
the reference's formulas on synthetic trajectories, and the HBM-resident batch assembly kernel (on the CPU emulation here, on
the GPU in tests/test_model_gpu.py)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from poseidon_amd import data as D

HERE = os.path.dirname(os.path.abspath(__file__))


def test_index_machinery_matches_reference_base_classes():
    cases = json.load(open(os.path.join(GOLDEN, "dataset_pins.json")))
    assert len(cases) == 72
    for c in cases:
        label = "[rho],[u,v],[p],[tracer]" if c["kind"] == "time" else "[u,v],[g]"
        traj, start, resolved = D.resolve_split(c["which"], c["num_trajectories"], 100, 12, 24)
        names, slices = D.channel_lists(label)
        assert (start, resolved, slices, names) == (c["start"], c["resolved_trajectories"], c["channel_slice_list"], c["descriptors"])
        assert label.count(",") + 1 == c["output_dim"]
        if c["kind"] == "steady":
            assert traj == c["length"]
            continue
        tp = D.TimePairs(**c["kw"])
        assert tp.multiplier == c["multiplier"] and traj * tp.multiplier == c["length"]
        assert [list(tp(i)) for i in c["idx"]] == c["maps"]
        i, t1, t2 = tp.arrays(np.array(c["idx"]))
        assert [[int(a), int(t2_ - t1_), int(t1_), int(t2_)] for a, t1_, t2_ in zip(i, t1, t2)] == c["maps"]


def _fake(key, n, T, C, R=16, seed=0):
    rng = np.random.default_rng(seed)
    return {key: rng.standard_normal((n, T, C, R, R)).astype(np.float32)}


def _mk(name, reader, **kw):
    ds = D.get_dataset(name, which="val", num_trajectories=3, reader=reader, n_max=12, n_val=4, n_test=3, **kw)
    ds.resolution = 16
    return ds


def test_incompressible_recipe_and_defaults():
    """fluids/incompressible.py:74-160: rho = 1 and p = 0 planes, (x - mean) / std with the shared constants, tracer channel 2 with
    its own constants, pixel mask on the pressure channel, time = dt / 20; `.out` = 10 time steps (base.py:118-121)."""
    rd = _fake("velocity", 12, 21, 3)
    ds = _mk("fluids.incompressible.PiecewiseConstants.tracer", rd)
    assert (ds.pairs.max_num_time_steps, ds.pairs.time_step_size) == (7, 2) and len(ds) == 4 * 36 and ds.start == 5
    assert ds.channel_slice_list == [0, 1, 3, 4, 5] and ds.pixel_mask.tolist() == [False, False, False, True, False] and ds.input_dim == 5
    idx = 2 * 36 + 17
    i, t, t1, t2 = ds.pairs(idx)
    s = ds[idx]
    v = rd["velocity"]
    mean, std = np.array([0.80, 0.0, 0.0, 0.0]), np.array([0.31, 0.391, 0.356, 0.185])
    for lab, tt in (("pixel_values", t1), ("labels", t2)):
        full = np.concatenate([np.ones((1, 16, 16)), v[i + 5, tt, 0:2], np.zeros((1, 16, 16))], 0)
        want = np.concatenate([(full - mean[:, None, None]) / std[:, None, None], (v[i + 5, tt, 2:3] - 0.19586183) / 0.37], 0)
        assert np.allclose(s[lab].numpy(), want, rtol=1e-6, atol=1e-6)
    assert s["time"] == pytest.approx(t / 20.0)
    assert D.get_dataset("fluids.incompressible.Sines.out", which="train", num_trajectories=2, reader=rd, n_max=12, n_val=4, n_test=3) \
        .pairs.max_num_time_steps == 10
    sl = _mk("fluids.incompressible.ShearLayer", rd)          # transposed fields (incompressible.py:104-106)
    i, t, t1, t2 = sl.pairs(3)
    assert np.allclose(sl[3]["pixel_values"][1].numpy(), (v[i + 5, t1, 0].T - 0.0) / 0.391, rtol=1e-6, atol=1e-6)
    with pytest.raises(ValueError):
        _mk("fluids.incompressible.Gaussians.tracer", rd)
    with pytest.raises(ValueError):
        _mk("wave.Layer", rd)


def test_compressible_recipes():
    """fluids/compressible.py:191-262 (pressure shifted by the dataset's mean pressure before normalisation), :114-188 (GCE-RT:
    channels 0:4 and 5, its own constants, time / 10, step size 1), :56-111 (CE-RM)."""
    rd = _fake("data", 12, 21, 5)
    ds = _mk("fluids.compressible.Riemann", rd)
    i, t, t1, t2 = ds.pairs(40)
    x = rd["data"][i + 5, t2, 0:4].copy()
    x[3] -= 0.215
    want = (x - np.array([0.80, 0.0, 0.0, 0.0])[:, None, None]) / np.array([0.31, 0.391, 0.356, 0.185])[:, None, None]
    assert np.allclose(ds[40]["labels"].numpy(), want, rtol=1e-6, atol=1e-6) and ds.channel_slice_list == [0, 1, 3, 4]
    assert _mk("fluids.compressible.RiemannKelvinHelmholtz", rd).spec.channels[3].shift == 1.33
    assert _mk("fluids.compressible.RiemannCurved", rd).spec.file == "/CE-CRP.nc"
    rt = _mk("fluids.compressible.gravity.RayleighTaylor", _fake("solution", 12, 11, 6))
    assert (rt.pairs.time_step_size, rt.spec.time_const, rt.channel_slice_list) == (1, 10.0, [0, 1, 3, 4, 5])
    assert [c.src for c in rt.spec.channels] == [0, 1, 2, 3, 5]
    with pytest.raises(ValueError):
        _mk("fluids.compressible.gravity.RayleighTaylor", _fake("solution", 12, 11, 6), max_num_time_steps=6, time_step_size=2)


@pytest.mark.parametrize("name,key,C", [("fluids.incompressible.PiecewiseConstants.tracer", "velocity", 3),
                                        ("fluids.incompressible.ShearLayer", "velocity", 2), ("fluids.compressible.Riemann", "data", 5)])
def test_device_batch_matches_getitem(name, key, C, monkeypatch):
    """DeviceTrajectories.batch (scot_gather_pairs, one launch) == the collated __getitem__ samples."""
    sys.path.insert(0, os.path.join(HERE, "hipemu"))
    import emu_session
    emu_session.patch_ops(monkeypatch, emu_session.load_emu())
    ds = _mk(name, _fake(key, 12, 21, C))
    dev = ds.to_device("cpu")
    idx = [0, 5, 36, 71, 143, 100]
    got = dev.batch(idx)
    for k, j in enumerate(idx):
        s = ds[j]
        assert np.allclose(got["pixel_values"][k].numpy(), s["pixel_values"].numpy(), rtol=1e-6, atol=1e-6)
        assert np.allclose(got["labels"][k].numpy(), s["labels"].numpy(), rtol=1e-6, atol=1e-6)
        assert float(got["time"][k]) == pytest.approx(s["time"]) and got["pixel_mask"][k].tolist() == s["pixel_mask"].tolist()
    with pytest.raises(IndexError):
        dev.batch([len(dev)])
