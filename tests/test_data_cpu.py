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
    # just_velocities (incompressible.py:44-63): the two velocity channels only, "[u,v]", mask [F, F]
    jv = _mk("fluids.incompressible.Sines", rd, just_velocities=True)
    i, t, t1, t2 = jv.pairs(7)
    assert jv.input_dim == 2 and jv.channel_slice_list == [0, 2] and jv.pixel_mask.tolist() == [False, False]
    assert np.allclose(jv[7]["labels"].numpy(), v[i + 5, t2, 0:2] / np.array([0.391, 0.356])[:, None, None], rtol=1e-6, atol=1e-6)
    with pytest.raises(ValueError):
        _mk("fluids.compressible.gravity.Blast", rd)      # named by the reference's registry, but no such reader exists there
    with pytest.raises(ValueError):
        _mk("nonsense.Dataset", rd)


def test_incompressible_resolution_option():
    """`resolution=` (incompressible.py:66-83, 141-143): samples spectrally downsampled by the readers' own fft2 -> crop -> ifft2."""
    rng = np.random.default_rng(3)
    rd = {"velocity": rng.standard_normal((12, 21, 2, 128, 128)).astype(np.float32)}
    ds = D.get_dataset("fluids.incompressible.Sines", which="val", num_trajectories=3, reader=rd, n_max=12, n_val=4, n_test=3, resolution=64)
    s = ds[5]
    assert s["pixel_values"].shape == (4, 64, 64) and s["labels"].shape == (4, 64, 64)
    i, t, t1, t2 = ds.pairs(5)
    full = torch.from_numpy(rd["velocity"][i + 5, t1, 0]) / 0.391
    assert float(s["pixel_values"][1].mean()) == pytest.approx(float(full.mean()), abs=1e-6)      # the mean (k = 0) survives the crop
    assert float(s["pixel_values"][0].std()) < 1e-6                                                 # a constant plane stays constant
    with pytest.raises(ValueError):
        D.get_dataset("fluids.incompressible.Sines", which="val", num_trajectories=3, reader=rd, n_max=12, n_val=4, n_test=3, resolution=256)


def test_forced_and_steady_fluids_recipes():
    """fluids/incompressible.py:149-243 (KolmogorovFlow: own velocity constants, analytic forcing channel in inputs AND labels) and
    fluids/compressible.py:8-53 (Airfoil: element-wise mask, label 1 inside the body, no time)."""
    rng = np.random.default_rng(4)
    rd = {"solution": rng.standard_normal((12, 21, 2, 128, 128)).astype(np.float32)}
    kw = dict(which="val", num_trajectories=3, n_max=12, n_val=4, n_test=3)
    ds = D.get_dataset("fluids.incompressible.forcing.KolmogorovFlow", reader=rd, **kw)
    assert ds.channel_slice_list == [0, 1, 3, 4, 5] and ds.pixel_mask.tolist() == [False, False, False, True, False] and ds.input_dim == 5
    i, t, t1, t2 = ds.pairs(40)
    s = ds[40]
    xs = torch.linspace(0, 1, 128)
    X, Y = torch.meshgrid(xs, xs, indexing="ij")
    forcing = ((0.1 * torch.sin(2.0 * np.pi * (X + Y))) - (-1.2996679288335145e-09)) / 0.0707106739282608
    assert np.allclose(s["pixel_values"][4].numpy(), forcing.numpy(), atol=2e-6) and torch.equal(s["labels"][4], s["pixel_values"][4])
    assert np.allclose(s["labels"][1].numpy(), (rd["solution"][i + 5, t2, 0] - (-2.2424793e-13)) / 0.22017328, rtol=1e-6, atol=1e-6)
    assert np.allclose(s["labels"][2].numpy(), (rd["solution"][i + 5, t2, 1] - 4.1510376e-12) / 0.22078253, rtol=1e-6, atol=1e-6)
    assert np.allclose(s["pixel_values"][0].numpy(), (1 - 0.80) / 0.31) and s["time"] == pytest.approx(t / 20.0)
    jv = D.get_dataset("fluids.incompressible.forcing.KolmogorovFlow", reader=rd, just_velocities=True, **kw)
    assert jv.label_description == "[u,v],[g]" and jv.input_dim == 3 and jv.pixel_mask.tolist() == [False, False, False]
    body = (rng.random((12, 1, 128, 128)) > 0.7).astype(np.float32)
    af = D.get_dataset("fluids.compressible.steady.Airfoil", reader={"solution": np.concatenate([body, rng.standard_normal((12, 1, 128, 128)).astype(np.float32)], 1)}, **kw)
    s = af[2]
    assert len(af) == 4 and set(s) == {"pixel_values", "labels", "pixel_mask"} and s["pixel_mask"].shape == (1, 128, 128)
    want = (af.reader["solution"][2 + 5, 1] - 0.92984116) / 0.10864315
    want = np.where(body[2 + 5, 0] == 1, 1.0, want)
    assert np.array_equal(s["pixel_values"][0].numpy(), body[2 + 5, 0]) and np.allclose(s["labels"][0].numpy(), want, rtol=1e-6, atol=1e-6)
    with pytest.raises(ValueError):
        D.get_dataset("fluids.compressible.steady.Airfoil.out", reader={"solution": body}, **kw)


def test_wave_reaction_diffusion_and_elliptic_recipes():
    """wave/acoustic.py (static wave speed c is input channel 1 AND label channel 1), reaction_diffusion/allen_cahn.py, elliptic/poisson.py,
    elliptic/helmholtz.py (inputs [a - 1, bc], per-sample groups), `.time` wrapper (base.py:372-395), defaults of base.py:121-157."""
    rng = np.random.default_rng(5)
    R, kw = 16, dict(which="val", num_trajectories=3, n_max=12, n_val=4, n_test=3)

    def mk(name, reader, **k2):
        ds = D.get_dataset(name, reader=reader, **kw, **k2)
        ds.resolution = R
        return ds
    wave = {"solution": rng.standard_normal((12, 21, R, R)).astype(np.float32), "c": (3000 + 500 * rng.standard_normal((12, R, R))).astype(np.float32)}
    ds = mk("wave.Layer", wave)
    i, t, t1, t2 = ds.pairs(50)
    s = ds[50]
    assert set(s) == {"pixel_values", "labels", "time"} and ds.channel_slice_list == [0, 1, 2] and ds.output_dim == 2
    c = (wave["c"][i + 5] - 3498.5644380917424) / 647.843958567462
    assert np.allclose(s["pixel_values"][0].numpy(), (wave["solution"][i + 5, t1] - 0.03467443221585092) / 0.10442421752963911, rtol=1e-6, atol=1e-6)
    assert np.allclose(s["labels"][0].numpy(), (wave["solution"][i + 5, t2] - 0.03467443221585092) / 0.10442421752963911, rtol=1e-6, atol=1e-6)
    assert np.allclose(s["pixel_values"][1].numpy(), c, rtol=1e-5, atol=1e-5) and torch.equal(s["labels"][1], s["pixel_values"][1])
    assert s["time"] == pytest.approx(t / 20.0) and mk("wave.Layer.out", wave).pairs.max_num_time_steps == 10
    g = mk("wave.Gaussians", wave)
    assert g.spec.time_const == 15.0 and g.spec.channels[1].mean == 2618.4593933 and g.spec.file == "/Wave-Gauss.nc"
    with pytest.raises(ValueError):
        mk("wave.Gaussians.out", wave)
    with pytest.raises(ValueError):
        mk("wave.Gaussians", wave, max_num_time_steps=8, time_step_size=2)     # 16 > 15 (acoustic.py:69)
    ac = mk("reaction_diffusion.AllenCahn", {"solution": wave["solution"][:, :20]})
    i, t, t1, t2 = ac.pairs(9)
    assert np.allclose(ac[9]["labels"][0].numpy(), (wave["solution"][i + 5, t2] - 0.002484262) / 0.65351176, rtol=1e-6, atol=1e-6)
    assert ac[9]["time"] == pytest.approx(t / 19.0) and mk("reaction_diffusion.AllenCahn.out", {"solution": wave["solution"]}).pairs.max_num_time_steps == 9
    po = mk("elliptic.poisson.Gaussians", {"source": wave["c"], "solution": wave["solution"][:, 0]})
    s = po[1]
    assert len(po) == 4 and set(s) == {"pixel_values", "labels"}
    assert np.allclose(s["pixel_values"][0].numpy(), (wave["c"][1 + 5] - 0.014822142414492256) / 4.755138816607612, rtol=1e-6, atol=1e-6)
    assert np.allclose(s["labels"][0].numpy(), (wave["solution"][1 + 5, 0] - 0.0005603458434937093) / 0.02401226126952699, rtol=1e-6, atol=1e-6)
    assert mk("elliptic.poisson.Gaussians.time", {"source": wave["c"], "solution": wave["solution"][:, 0]})[1]["time"] == 1.0
    bc = rng.standard_normal(12).astype(np.float32)
    groups = {f"Sample_{j}": {"a": wave["c"][j], "bc": np.float32(bc[j]), "u": wave["solution"][j, 3]} for j in range(12)}   # the file's layout
    for reader in (groups, {"a": wave["c"], "bc": bc, "u": wave["solution"][:, 3]}):
        he = mk("elliptic.Helmholtz.time", reader)
        s = he[2]
        assert np.allclose(s["pixel_values"][0].numpy(), wave["c"][2 + 5] - 1, rtol=1e-6) and np.allclose(s["pixel_values"][1].numpy(), bc[2 + 5])
        assert np.allclose(s["labels"][0].numpy(), (wave["solution"][2 + 5, 3] - 0.11523915668552) / 0.8279975746000605, rtol=1e-6, atol=1e-6)
        assert s["time"] == 1.0 and he.input_dim == 2 and he.output_dim == 1
    with pytest.raises(NotImplementedError):
        mk("elliptic.Helmholtz.out", groups)


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


def _family_reader(name, R):
    rng = np.random.default_rng(11)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    if name.startswith("wave"):
        return {"solution": f(12, 21, R, R), "c": f(12, R, R)}
    if "AllenCahn" in name:
        return {"solution": f(12, 20, R, R)}
    if "poisson" in name:
        return {"source": f(12, R, R), "solution": f(12, R, R)}
    if "Helmholtz" in name:
        return {"a": f(12, R, R), "bc": f(12), "u": f(12, R, R)}
    if "Airfoil" in name:
        return {"solution": np.concatenate([(rng.random((12, 1, R, R)) > 0.6).astype(np.float32), f(12, 1, R, R)], 1)}
    return {"solution": f(12, 21, 2, R, R)}


@pytest.mark.parametrize("name", ["wave.Layer", "reaction_diffusion.AllenCahn", "elliptic.poisson.Gaussians", "elliptic.Helmholtz.time",
                                  "fluids.compressible.steady.Airfoil", "fluids.incompressible.forcing.KolmogorovFlow"])
def test_device_batch_matches_getitem_other_families(name, monkeypatch):
    """DeviceTrajectories.batch through scot_gather_planes (one launch per tensor: inputs and labels follow different recipes, static
    / analytic / scalar source planes) == the collated __getitem__ samples, for every non-fluids family."""
    sys.path.insert(0, os.path.join(HERE, "hipemu"))
    import emu_session
    emu_session.patch_ops(monkeypatch, emu_session.load_emu())
    R = 128 if "Kolmogorov" in name else 16
    ds = D.get_dataset(name, which="val", num_trajectories=3, reader=_family_reader(name, R), n_max=12, n_val=4, n_test=3)
    ds.resolution = R
    dev = ds.to_device("cpu")
    idx = [0, 3, 1] if ds.steady else [0, 5, 36, 71, 143, 100]
    got = dev.batch(idx)
    for k, j in enumerate(idx):
        s = ds[j]
        assert set(got) == set(s)
        assert np.allclose(got["pixel_values"][k].numpy(), s["pixel_values"].numpy(), rtol=1e-6, atol=1e-6)
        assert np.allclose(got["labels"][k].numpy(), s["labels"].numpy(), rtol=1e-6, atol=1e-6)
        if "time" in s:
            assert float(got["time"][k]) == pytest.approx(s["time"])
        if "pixel_mask" in s:
            assert torch.equal(got["pixel_mask"][k].cpu(), s["pixel_mask"])
    with pytest.raises(IndexError):
        dev.batch([len(dev)])


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


def test_reference_type_names_and_size_helpers():
    """What the reference's drivers import besides get_dataset: `BaseTimeDataset` for their `time_involved` isinstance checks
    (train.py:227-231, inference.py:74) and the two parameter counters of scOT/utils.py."""
    from scOT.problems.base import BaseDataset, BaseTimeDataset, get_dataset
    from scOT.utils import get_num_parameters, get_num_parameters_no_embed
    rng = np.random.default_rng(0)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    kw = dict(which="val", num_trajectories=3, n_max=12, n_val=4, n_test=3)
    wave = get_dataset("wave.Layer", reader={"solution": f(12, 21, 16, 16), "c": f(12, 16, 16)}, **kw)
    po = {"source": f(12, 16, 16), "solution": f(12, 16, 16)}
    steady, wrapped = get_dataset("elliptic.poisson.Gaussians", reader=po, **kw), get_dataset("elliptic.poisson.Gaussians.time", reader=po, **kw)
    assert all(isinstance(d, BaseDataset) for d in (wave, steady, wrapped))
    assert isinstance(wave, BaseTimeDataset) and isinstance(wrapped, BaseTimeDataset) and not isinstance(steady, BaseTimeDataset)
    m = torch.nn.ModuleDict({"embeddings": torch.nn.Linear(3, 2), "body": torch.nn.Linear(2, 2), "patch_recovery": torch.nn.Linear(2, 1)})
    m["body"].bias.requires_grad_(False)
    assert get_num_parameters(m) == 8 + 4 + 3 and get_num_parameters_no_embed(m) == 4


# ---------------------------------------------------------------------------------------------------------------------------
# recipes pinned to the reference's own readers (tests/golden/make_dataset_recipe_pins.py ran /root/reference/scOT/problems/** on
# tests/golden/synth_h5.py's synthetic files; the same files are regenerated here from their seed)
def _recipe_pins():
    import json
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dataset_recipe_pins.json")))


def _synth_reader(ds_name):
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import synth_h5
    spec, _ = D._spec(ds_name.replace(".time", ""))
    return synth_h5, synth_h5.SynthFile(spec.file)


def check_against_pin(pin, sample_of, tol=2e-6):
    """sample_of(dataset, idx) -> the sample dict (CPU reader or one row of a device batch)"""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import synth_h5
    if "raises" in pin:
        with pytest.raises((ValueError, AssertionError, TypeError, NotImplementedError)):
            D.get_dataset(pin["name"], which=pin["which"], num_trajectories=pin["num_trajectories"], reader={}, **pin["kw"])
        return
    _, reader = _synth_reader(pin["name"])
    ds = D.get_dataset(pin["name"], which=pin["which"], num_trajectories=pin["num_trajectories"], reader=reader, **pin["kw"])
    assert len(ds) == pin["length"] and ds.input_dim == pin["input_dim"] and ds.output_dim == pin["output_dim"]
    assert list(ds.channel_slice_list) == pin["channel_slice_list"] and ds.start == pin["start"]
    for idx, want in zip(pin["idx"], pin["samples"]):
        got = sample_of(ds, idx)
        assert set(got) == set(want), (pin["name"], set(got), set(want))
        for k, w in want.items():
            if k in ("pixel_values", "labels"):
                g = synth_h5.summary(got[k].detach().cpu().numpy())
                assert g["shape"] == w["shape"], (pin["name"], k)
                n = float(np.prod(w["shape"]))
                for m in ("sum", "abs", "wsum"):
                    assert abs(g[m] - w[m]) <= tol * (w["abs"] + n) * (16 if m == "wsum" else 1), (pin["name"], pin["kw"], k, m, g[m], w[m])
                np.testing.assert_allclose(g["sub"], w["sub"], rtol=1e-5, atol=1e-5, err_msg=f"{pin['name']} {k}")
            elif k == "pixel_mask":
                m = got[k].detach().cpu().numpy()
                assert list(m.shape) == w["shape"] and int(m.sum()) == w["count"] and str(got[k].dtype) == w["dtype"]
                if w["values"] is not None:
                    assert m.astype(int).tolist() == w["values"]
                else:
                    assert abs(synth_h5.summary(m.astype(np.float32))["wsum"] - w["wsum"]) < 1e-6
            else:
                assert abs(float(got[k]) - w) < 1e-7, (pin["name"], k)


@pytest.mark.parametrize("pin", _recipe_pins(), ids=lambda p: f"{p['name']}-{p['which']}" + ("-" + "_".join(p["kw"]) if p["kw"] else ""))
def test_recipes_match_the_reference_readers(pin):
    check_against_pin(pin, lambda ds, i: ds[i])


def device_rows(ds, idx, device):
    """one row of a DeviceTrajectories batch in the shape of a __getitem__ sample"""
    b = ds.to_device(device).batch([idx])
    out = {}
    for k, v in b.items():
        out[k] = v[0] if k != "time" else float(v[0])
    return out


_DEVICE_PIN_NAMES = ("fluids.incompressible.ShearLayer", "fluids.compressible.steady.Airfoil", "wave.Gaussians", "elliptic.Helmholtz.time",
                     "fluids.incompressible.forcing.KolmogorovFlow")      # (the emulator runs one host thread per lane: the GPU test covers every pin)


@pytest.mark.parametrize("pin", [p for p in _recipe_pins() if p["name"] in _DEVICE_PIN_NAMES and p["which"] == "test" and not p["kw"]],
                         ids=lambda p: p["name"])
def test_device_batches_match_the_reference_readers_emulated(pin, monkeypatch):
    """the HBM-resident twin (scot_gather_pairs / scot_gather_planes on the CPU emulation) against the same pins"""
    sys.path.insert(0, os.path.join(HERE, "hipemu"))
    import emu_session
    emu_session.patch_ops(monkeypatch, emu_session.load_emu())
    check_against_pin(pin, lambda ds, i: device_rows(ds, i, "cpu"))


def test_numpy_file_readers(tmp_path):
    """`.npy` (memory-mapped), `.npz` and per-array `.npy` directories stand in for the HDF5 files: get_dataset(data_path=...) finds an
    export lying beside the name the reference opens (fluids/incompressible.py:36-38) and yields the same samples as the arrays."""
    rng = np.random.default_rng(5)
    vel = rng.standard_normal((12, 21, 3, 128, 128)).astype(np.float32)
    kw = dict(which="val", num_trajectories=3, n_max=12, n_val=4, n_test=3)
    want = D.get_dataset("fluids.incompressible.Sines", reader={"velocity": vel}, **kw)[17]
    a, b, c = tmp_path / "a", tmp_path / "b", tmp_path / "c"
    for d in (a, b, c):
        d.mkdir()
    np.save(a / "NS-Sines.npy", vel)
    np.savez(b / "NS-Sines.npz", velocity=vel)
    D.export_npy({"velocity": vel}, ["velocity"], str(c / "NS-Sines"))
    for d in (a, b, c):
        ds = D.get_dataset("fluids.incompressible.Sines", data_path=str(d), **kw)
        got = ds[17]
        assert torch.equal(got["pixel_values"], want["pixel_values"]) and torch.equal(got["labels"], want["labels"]) and got["time"] == want["time"]
    assert isinstance(D.open_reader(str(a / "NS-Sines.nc"))["velocity"], np.memmap)
    assert isinstance(D.open_reader(str(c / "NS-Sines.nc"))["velocity"], np.memmap)
    wave = {"solution": rng.standard_normal((12, 21, 128, 128)).astype(np.float32), "c": rng.standard_normal((12, 128, 128)).astype(np.float32)}
    D.export_npy(wave, ["solution", "c"], str(c / "Wave-Layer"))
    w0, w1 = D.get_dataset("wave.Layer", reader=wave, **kw)[5], D.get_dataset("wave.Layer", data_path=str(c), **kw)[5]
    assert torch.equal(w0["pixel_values"], w1["pixel_values"]) and torch.equal(w0["labels"], w1["labels"])
    with pytest.raises((ImportError, OSError)):
        D.get_dataset("fluids.incompressible.Gaussians", data_path=str(c), **kw)     # nothing exported under that name, no h5py
