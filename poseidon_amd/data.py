"""Dataset front-end for the scOT hot path: the reference's index machinery and per-dataset sample
recipes, with the trajectories RESIDENT IN HBM and a batch assembled by one HIP kernel instead of a CPU DataLoader.

What is mirrored (reference `scOT/problems/base.py`, `scOT/problems/fluids/*.py`):
  * splits — train / val / test from (N_max, N_val, N_test), `num_trajectories` -1 / -2 / -8 = all / half / an eighth
    (base.py:222-247, 336-369);
  * time pairs — every (t1 <= t2) on the `time_step_size` grid up to `max_num_time_steps`, optionally restricted to
    `allowed_time_transitions` or anchored at `fix_input_to_time_step`; `idx -> (trajectory, t2 - t1, t1, t2)` (base.py:318-334);
  * channel groups of the loss from the label description "[rho],[u,v],[p]" -> [0, 1, 3, 4] (base.py:272-285);
  * `get_dataset(name)` — name -> dataset + default time settings (".out" variants: 10 steps), ".tracer" (base.py:15-160);
  * per-dataset sample = channels of one array at t1 / t2, constant planes (incompressible density 1 / pressure 0), per-channel
    (x - mean) / std, a mean-pressure shift, transposition (shear layer), time = (t2 - t1) / T (fluids/incompressible.py:74-160,
    fluids/compressible.py:56-262).  Covered: the fluids family (NS-*, CE-*, GCE-RT, CE-RM) — the datasets of BASELINE.json's
    configs; the wave / elliptic / reaction-diffusion / forced-NS / airfoil readers are not restated (ValueError).
The index machinery is pinned against the reference's own base classes (tests/golden/make_dataset_pins.py); the per-dataset
recipes are restated from the reference's readers, which need h5py to import and are therefore unpinned here (tests compare the
HIP batch with a numpy evaluation of the recipe).

MI355X-native part: `DeviceTrajectories` keeps a whole dataset in HBM (CE-RP: 10000 x 21 x 5 x 128^2 fp32 = 69 GB of 288 GB; a
training subset far less) and `batch(indices)` gathers + normalises `pixel_values` / `labels` for a batch with ONE launch
(`scot_gather_pairs`), so the input pipeline never touches the host.  `PDEDataset.__getitem__` is the CPU path with the
reference's dict (`pixel_values`, `labels`, `time`, `pixel_mask`) for torch DataLoader users.
Readers: anything indexable like `reader[key][i, t, c0:c1]` — a dict of numpy arrays / memmaps, or an `h5py.File` where h5py
is installed (it is not in this image; `open_reader` raises a clear error then).
"""
from __future__ import annotations

import os
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

# ------------------------------------------------------------------------------------------------ index machinery


def channel_lists(label_description: str) -> Tuple[List[str], List[int]]:
    """"[rho],[u,v],[p]" -> (["rho", "uv", "p"], [0, 1, 3, 4])   (reference base.py:272-285)."""
    groups = re.findall(r"\[([^\[\]]+)\]", label_description)
    slices, names = [0], []
    for g in groups:
        parts = g.split(",")
        slices.append(slices[-1] + len(parts))
        names.append("".join(parts) if len(parts) > 1 else g)
    return names, slices


def resolve_split(which: str, num_trajectories: int, n_max: int, n_val: int, n_test: int) -> Tuple[int, int, int]:
    """-> (trajectories in this split, first trajectory, resolved training trajectories)   (reference base.py:222-247)."""
    if which not in ("train", "val", "test"):
        raise ValueError("which must be train, val or test")
    if not (num_trajectories is not None and (num_trajectories > 0 or num_trajectories in (-1, -2, -8))):
        raise ValueError("num_trajectories must be positive or one of -1, -2, -8")
    if not (n_max > 0 and n_max >= n_val + n_test and n_val > 0 and n_test > 0):
        raise ValueError("inconsistent dataset sizes")
    free = n_max - n_val - n_test
    nt = {-1: free, -2: free // 2, -8: free // 8}.get(num_trajectories, num_trajectories)
    if nt + n_val + n_test > n_max:
        raise ValueError("num_trajectories exceeds the training part of the dataset")
    if which == "train":
        return nt, 0, nt
    if which == "val":
        return n_val, n_max - n_val - n_test, nt
    return n_test, n_max - n_test, nt


class TimePairs:
    """idx -> (trajectory, dt, t1, t2) of a time-dependent dataset (reference base.py:318-334, 356-369)."""

    def __init__(self, max_num_time_steps: int, time_step_size: int, fix_input_to_time_step: Optional[int] = None,
                 allowed_time_transitions: Optional[Sequence[int]] = None):
        if not (max_num_time_steps and max_num_time_steps > 0 and time_step_size and time_step_size > 0):
            raise ValueError("max_num_time_steps and time_step_size must be positive")
        if fix_input_to_time_step is not None and fix_input_to_time_step < 0:
            raise ValueError("fix_input_to_time_step must be >= 0")
        self.max_num_time_steps, self.time_step_size = max_num_time_steps, time_step_size
        self.fix = fix_input_to_time_step
        if self.fix is not None:
            self.pairs = None
            self.multiplier = max_num_time_steps
        else:
            self.pairs = [(time_step_size * i, time_step_size * j) for i in range(max_num_time_steps + 1)
                          for j in range(i, max_num_time_steps + 1)
                          if allowed_time_transitions is None or (j - i) in allowed_time_transitions]
            self.multiplier = len(self.pairs)

    def __call__(self, idx: int) -> Tuple[int, int, int, int]:
        i, r = divmod(idx, self.multiplier)
        if self.fix is None:
            t1, t2 = self.pairs[r]
        else:
            t1, t2 = self.fix, self.time_step_size * (r + 1) + self.fix
        return i, t2 - t1, t1, t2

    def arrays(self, idx: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """vectorised: (trajectory, t1, t2) for an array of sample indices"""
        i, r = np.divmod(np.asarray(idx, dtype=np.int64), self.multiplier)
        if self.fix is None:
            p = np.asarray(self.pairs, dtype=np.int64)
            return i, p[r, 0], p[r, 1]
        return i, np.full_like(r, self.fix), self.time_step_size * (r + 1) + self.fix


# ------------------------------------------------------------------------------------------------ per-dataset recipes
@dataclass
class Channel:
    """One output channel: `(plane - mean) / std` with plane = array channel `src` (minus `shift`) or the constant `const`."""
    src: Optional[int] = None
    const: float = 0.0
    mean: float = 0.0
    std: float = 1.0
    shift: float = 0.0

    def affine(self) -> Tuple[float, float]:
        """out = a * x + b   (x = the array value; constant channels: a = 0)"""
        if self.src is None:
            return 0.0, (self.const - self.mean) / self.std
        return 1.0 / self.std, -(self.shift + self.mean) / self.std


@dataclass
class DatasetSpec:
    file: str
    key: str
    n_max: int
    n_val: int
    n_test: int
    channels: List[Channel]
    label_description: str
    pixel_mask: List[bool]
    time_const: float
    max_time_index: int            # max_num_time_steps * time_step_size must not exceed this (the readers' asserts)
    resolution: int = 128
    transpose: bool = False
    defaults: Dict[str, int] = field(default_factory=lambda: dict(max_num_time_steps=7, time_step_size=2))


# reference scOT/problems/fluids/normalization_constants.py (mean / std of [rho, u, v, p], tracer)
_NS = dict(mean=[0.80, 0.0, 0.0, 0.0], std=[0.31, 0.391, 0.356, 0.185], tracer_mean=0.19586183, tracer_std=0.37, time=20.0)


def _incompressible(file: str, n_max: int, tracer: bool, transpose: bool = False) -> DatasetSpec:
    """fluids/incompressible.py:9-160: velocity from the file, density = 1 and pressure = 0 planes, optional passive tracer."""
    m, s = _NS["mean"], _NS["std"]
    ch = [Channel(const=1.0, mean=m[0], std=s[0]), Channel(src=0, mean=m[1], std=s[1]), Channel(src=1, mean=m[2], std=s[2]),
          Channel(const=0.0, mean=m[3], std=s[3])]
    desc, mask = "[rho],[u,v],[p]", [False, False, False, True]
    if tracer:
        ch.append(Channel(src=2, mean=_NS["tracer_mean"], std=_NS["tracer_std"]))
        desc, mask = desc + ",[tracer]", mask + [False]
    return DatasetSpec(file, "velocity", n_max, 120, 240, ch, desc, mask, _NS["time"], 20, transpose=transpose)


def _compressible(file: str, mean_pressure: float) -> DatasetSpec:
    """fluids/compressible.py:191-262: [rho, u, v, p] from `data`, pressure shifted by the dataset's mean pressure first."""
    m, s = _NS["mean"], _NS["std"]
    ch = [Channel(src=k, mean=m[k], std=s[k], shift=mean_pressure if k == 3 else 0.0) for k in range(4)]
    return DatasetSpec(file, "data", 10000, 120, 240, ch, "[rho],[u,v],[p]", [False] * 4, _NS["time"], 20)


def _spec(name: str) -> Tuple[DatasetSpec, Dict[str, int]]:
    """name -> (recipe, default time settings)   (reference base.py:15-160)"""
    tracer = "tracer" in name
    out = "out" in name          # (the reference tests the substring, base.py:80,118)
    if "fluids.incompressible" in name:
        table = {"BrownianBridge": ("/NS-BB.nc", 20000, False), "Gaussians": ("/NS-Gauss.nc", 20000, False),
                 "ShearLayer": ("/NS-SL.nc", 40000, False), "Sines": ("/NS-Sines.nc", 20000, False),
                 "PiecewiseConstants": ("/NS-PwC.nc", 20000, True), "VortexSheet": ("/NS-SVS.nc", 20000, False)}
        for key, (file, n_max, has_tracer) in table.items():
            if key in name:
                if tracer and not has_tracer:
                    raise ValueError(f"{key} does not have a tracer")
                spec = _incompressible(file, n_max, tracer, transpose=key == "ShearLayer")
                break
        else:
            raise ValueError(f"Unknown dataset {name}")
        dflt = dict(max_num_time_steps=10 if out else 7, time_step_size=2)
    elif "fluids.compressible" in name:
        if "gravity" in name and "RayleighTaylor" in name:
            # fluids/compressible.py:114-188: channels 0:4 and 5 (gravitational potential) of `solution`
            mean = [0.8970493, 4.0316996e-13, -1.3858967e-13, 0.7133829, -1.7055787]
            std = [0.12857835, 0.014896976, 0.014896975, 0.21293919, 0.40131348]
            ch = [Channel(src=k, mean=mean[k], std=std[k]) for k in range(4)] + [Channel(src=5, mean=mean[4], std=std[4])]
            spec = DatasetSpec("/GCE-RT.nc", "solution", 1260, 100, 130, ch, "[rho],[u,v],[p],[g]", [False] * 5, 10.0, 10)
            dflt = dict(max_num_time_steps=10 if out else 7, time_step_size=1)
        elif "RichtmyerMeshkov" in name:
            mean = [1.1964245, -7.164812e-06, 2.8968952e-06, 1.5648036]
            std = [0.5543239, 0.24304213, 0.2430597, 0.89639103]
            ch = [Channel(src=k, mean=mean[k], std=std[k]) for k in range(4)]
            spec = DatasetSpec("/CE-RM.nc", "solution", 1260, 100, 130, ch, "[rho],[u,v],[p]", [False] * 4, 20.0, 20)
            dflt = dict(max_num_time_steps=10 if out else 7, time_step_size=2)
        else:
            table = [("RiemannKelvinHelmholtz", "/CE-RPUI.nc", 1.33), ("RiemannCurved", "/CE-CRP.nc", 0.553), ("Riemann", "/CE-RP.nc", 0.215),
                     ("KelvinHelmholtz", "/CE-KH.nc", 1.0), ("Gaussians", "/CE-Gauss.nc", 2.513)]
            for key, file, mp in table:
                if key in name:
                    if tracer:
                        raise NotImplementedError(f"Tracer not implemented for {key}")
                    spec = _compressible(file, mp)
                    break
            else:
                raise ValueError(f"Unknown dataset {name}")
            dflt = dict(max_num_time_steps=10 if out else 7, time_step_size=2)
    else:
        raise ValueError(f"Unknown dataset {name} (this front-end restates the fluids family only)")
    return spec, dflt


def open_reader(path: str):
    """`.nc` / `.h5` files need h5py (HDF5 / netCDF-4), `.npz` / `.npy` work with numpy alone."""
    if path.endswith((".npz", ".npy")):
        r = np.load(path, mmap_mode="r")
        return r if path.endswith(".npz") else {"data": r, "solution": r, "velocity": r}
    try:
        import h5py
    except ImportError as e:   # pragma: no cover - environment dependent
        raise ImportError(f"reading {path} needs h5py (not installed here); pass reader={{key: array}} or a .npz/.npy file") from e
    return h5py.File(path, "r")


# ------------------------------------------------------------------------------------------------ datasets
class PDEDataset(torch.utils.data.Dataset):
    """A time-dependent fluids dataset with the reference's sample dict; `to_device()` gives the HBM-resident twin."""

    def __init__(self, name: str, which: str = "train", num_trajectories: int = -1, data_path: str = "./data", reader=None,
                 max_num_time_steps: Optional[int] = None, time_step_size: Optional[int] = None,
                 fix_input_to_time_step: Optional[int] = None, allowed_time_transitions: Optional[Sequence[int]] = None,
                 n_max: Optional[int] = None, n_val: Optional[int] = None, n_test: Optional[int] = None, **_):
        spec, dflt = _spec(name)
        self.name, self.spec, self.which = name, spec, which
        steps = max_num_time_steps if max_num_time_steps is not None else dflt["max_num_time_steps"]
        dt = time_step_size if time_step_size is not None else dflt["time_step_size"]
        if steps * dt > spec.max_time_index:
            raise ValueError(f"max_num_time_steps * time_step_size must be <= {spec.max_time_index} for {name}")
        self.pairs = TimePairs(steps, dt, fix_input_to_time_step, allowed_time_transitions)
        # (n_max / n_val / n_test overrides: subsets of a dataset, e.g. for tests; defaults = the reference's sizes)
        self.n_max, self.n_val, self.n_test = n_max or spec.n_max, n_val or spec.n_val, n_test or spec.n_test
        self.trajectories, self.start, self.num_trajectories = resolve_split(which, num_trajectories, self.n_max, self.n_val, self.n_test)
        self.length = self.trajectories * self.pairs.multiplier
        self.resolution = spec.resolution
        self.input_dim = len(spec.channels)
        self.label_description = spec.label_description
        self.output_dim = spec.label_description.count(",") + 1
        self.printable_channel_description, self.channel_slice_list = channel_lists(spec.label_description)
        self.pixel_mask = torch.tensor(spec.pixel_mask)
        self.reader = reader if reader is not None else open_reader(os.path.join(data_path, spec.file.lstrip("/")))

    def __len__(self) -> int:
        return self.length

    def _planes(self, i: int, t: int) -> torch.Tensor:
        arr = self.reader[self.spec.key]
        out = []
        for c in self.spec.channels:
            if c.src is None:
                x = torch.full((self.resolution, self.resolution), float(c.const), dtype=torch.float32)
            else:
                x = torch.from_numpy(np.asarray(arr[i + self.start, t, c.src:c.src + 1])).type(torch.float32)
                x = x.reshape(self.resolution, self.resolution)
                if self.spec.transpose:
                    x = x.transpose(-2, -1)
                if c.shift:
                    x = x - c.shift
            out.append((x - c.mean) / c.std)
        return torch.stack(out, 0)

    def __getitem__(self, idx: int) -> Dict:
        i, t, t1, t2 = self.pairs(idx)
        return {"pixel_values": self._planes(i, t1), "labels": self._planes(i, t2), "time": t / self.spec.time_const,
                "pixel_mask": self.pixel_mask}

    def to_device(self, device="cuda", trajectories: Optional[int] = None) -> "DeviceTrajectories":
        return DeviceTrajectories(self, device, trajectories)


class DeviceTrajectories:
    """This split's trajectories resident in HBM; `batch(indices)` -> the collated dict of a reference batch in one launch."""

    def __init__(self, ds: PDEDataset, device="cuda", trajectories: Optional[int] = None):
        self.ds = ds
        n = ds.trajectories if trajectories is None else min(trajectories, ds.trajectories)
        arr = ds.reader[ds.spec.key]
        self.nsrc = max(c.src for c in ds.spec.channels if c.src is not None) + 1
        host = np.ascontiguousarray(np.asarray(arr[ds.start:ds.start + n, :, 0:self.nsrc], dtype=np.float32))
        self.data = torch.from_numpy(host).to(device)                       # [n, T, nsrc, H, W]
        self.n, self.T = n, host.shape[1]
        ab = [c.affine() for c in ds.spec.channels]
        self.src = torch.tensor([-1 if c.src is None else c.src for c in ds.spec.channels], dtype=torch.int32, device=device)
        self.a = torch.tensor([x[0] for x in ab], dtype=torch.float32, device=device)
        self.b = torch.tensor([x[1] for x in ab], dtype=torch.float32, device=device)
        self.pixel_mask = ds.pixel_mask.to(device)

    def __len__(self) -> int:
        return self.n * self.ds.pairs.multiplier

    def batch(self, indices) -> Dict[str, torch.Tensor]:
        from . import ops
        idx = np.asarray(indices, dtype=np.int64)
        i, t1, t2 = self.ds.pairs.arrays(idx)
        if idx.size == 0 or i.max() >= self.n or t2.max() >= self.T:
            raise IndexError("sample index outside the resident trajectories")
        dev = self.data.device
        it = torch.from_numpy(np.stack([i, t1, t2], 0).astype(np.int32)).to(dev)
        B, C, R = idx.size, len(self.ds.spec.channels), self.ds.resolution
        pv = torch.empty(B, C, R, R, dtype=torch.float32, device=dev)
        lab = torch.empty_like(pv)
        ops.gather_pairs(self.data, it, self.src, self.a, self.b, pv, lab, self.T, self.nsrc, R, R, bool(self.ds.spec.transpose))
        time = torch.from_numpy(((t2 - t1) / self.ds.spec.time_const).astype(np.float32)).to(dev)
        return {"pixel_values": pv, "labels": lab, "time": time, "pixel_mask": self.pixel_mask.unsqueeze(0).expand(B, -1)}


def get_dataset(dataset, **kwargs):
    """reference `get_dataset` (base.py:15-160) for the fluids family; a list of names -> torch ConcatDataset."""
    if isinstance(dataset, (list, tuple)):
        return torch.utils.data.ConcatDataset([get_dataset(d, **kwargs) for d in dataset])
    return PDEDataset(dataset, **kwargs)
