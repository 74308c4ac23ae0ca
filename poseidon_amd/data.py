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
    fluids/compressible.py:56-262).  Covered: every reader of the reference — the fluids family (NS-*, CE-*, GCE-RT, CE-RM: the
    datasets of BASELINE.json's configs), forced NS (FNS-KF: analytic forcing plane, `just_velocities`), the steady airfoil (SE-AF:
    element-wise pixel mask), wave (Wave-Layer / Wave-Gauss: a static wave-speed channel passed through to the labels), Allen-Cahn,
    Poisson and Helmholtz (time-independent; ".time" wraps them with time = 1.0, base.py:372-395).  `...gravity.Blast` is named by
    the reference's registry but has no reader there (ImportError in the reference): ValueError here.
The index machinery is pinned against the reference's own base classes (tests/golden/make_dataset_pins.py), and so are the
per-dataset recipes: tests/golden/make_dataset_recipe_pins.py runs the reference's readers on an h5py-shaped synthetic file source
and pins `get_dataset(name)[i]` for every registry name (138 pins, checked on the CPU readers, the emulated device path and the GPU).

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
    """One output channel: `(plane - mean) / std` with plane = a slice of array `key` (minus `shift`), the constant `const`, or a
    fixed `plane`.  layout — how array `key` is indexed for trajectory i at time index t:
      "ntc": arr[i, t, src]   (fluids: [n, T, C, H, W])        "nt": arr[i, t]   ([n, T, H, W]: wave, Allen-Cahn)
      "n":   arr[i]           ([n, H, W]: static in time)       "nk": arr[i, src] ([n, 2, H, W]: the airfoil's two planes)
      "s":   arr[i] scalar    (Helmholtz boundary value, broadcast over the plane)"""
    src: Optional[int] = None
    const: float = 0.0
    mean: float = 0.0
    std: float = 1.0
    shift: float = 0.0
    key: Optional[str] = None          # None: the spec's main array
    layout: str = "ntc"
    plane: Optional[np.ndarray] = None  # fixed [H, W] plane (already normalised): KolmogorovFlow's forcing

    def is_array(self) -> bool:
        return self.plane is None and (self.src is not None or self.layout in ("nt", "n", "s"))

    def affine(self) -> Tuple[float, float]:
        """out = a * x + b   (x = the array value / the fixed plane; constant channels: a = 0)"""
        if self.plane is not None:
            return 1.0, 0.0
        if not self.is_array():
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
    labels: Optional[List[Channel]] = None     # None: the input recipe evaluated at t2
    steady: bool = False                       # time-independent: one sample per trajectory, no "time" key (unless ".time"-wrapped)
    has_pixel_mask: bool = True                # the fluids readers return a per-channel pixel_mask; the others none
    input_mask_value: Optional[float] = None   # airfoil: pixel_mask = (input == value) element-wise, labels there set to it


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


def _spec(name: str, just_velocities: bool = False) -> Tuple[DatasetSpec, Dict[str, int]]:
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
            if "forcing" in name and "KolmogorovFlow" in name:
                if tracer:
                    raise ValueError("KolmogorovFlow does not have a tracer")
                spec = _kolmogorov(just_velocities)
            else:
                raise ValueError(f"Unknown dataset {name}")
        dflt = dict(max_num_time_steps=10 if out else 7, time_step_size=2)
    elif "fluids.compressible" in name:
        if "gravity" in name and "Blast" in name:
            raise ValueError(f"{name}: the reference's registry names fluids.compressible.Blast but ships no such reader")
        if "steady" in name:
            if "steady.Airfoil" not in name or out:
                raise ValueError(f"Unknown dataset {name}")
            return _airfoil(), dict(max_num_time_steps=1, time_step_size=1)
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
    elif "elliptic" in name:
        if ".out" in name:
            raise NotImplementedError(f"Unknown dataset {name}")
        if "elliptic.poisson" in name:
            if "Gaussians" not in name:
                raise ValueError(f"Unknown dataset {name}")
            # elliptic/poisson.py:15-50
            spec = DatasetSpec("/Poisson-Gauss.nc", "source", 20000, 120, 240,
                               [Channel(key="source", layout="n", mean=0.014822142414492256, std=4.755138816607612)], "[u]", [False], 1.0, 0,
                               labels=[Channel(key="solution", layout="n", mean=0.0005603458434937093, std=0.02401226126952699)],
                               steady=True, has_pixel_mask=False)
        elif "elliptic.Helmholtz" in name:
            # elliptic/helmholtz.py:9-49: inputs [a - 1, bc·1], labels (u - mean) / std; per-sample groups "Sample_<i>" in the file
            spec = DatasetSpec("/Helmholtz.h5", "a", 19675, 128, 512,
                               [Channel(key="a", layout="n", shift=1.0), Channel(key="bc", layout="s")], "[u]", [False], 1.0, 0,
                               labels=[Channel(key="u", layout="n", mean=0.11523915668552, std=0.8279975746000605)],
                               steady=True, has_pixel_mask=False)
        else:
            raise ValueError(f"Unknown dataset {name}")
        dflt = dict(max_num_time_steps=1, time_step_size=1)
    elif "wave" in name:
        # wave/acoustic.py:6-125: [u(t), c] -> [u(t'), c]; the wave speed c is static and is ALSO the second label channel
        if "wave.Layer" in name:
            k = dict(mean=0.03467443221585092, std=0.10442421752963911, mean_c=3498.5644380917424, std_c=647.843958567462, time=20.0)
            file, steps = "/Wave-Layer.nc", 10 if out else 7
        elif "wave.Gaussians" in name:
            if out:
                raise ValueError(f"Unknown dataset {name}")
            k = dict(mean=0.0334376316, std=0.1171879068, mean_c=2618.4593933, std_c=601.51658913, time=15.0)
            file, steps = "/Wave-Gauss.nc", 7
        else:
            raise ValueError(f"Unknown dataset {name}")
        ch = [Channel(key="solution", layout="nt", mean=k["mean"], std=k["std"]), Channel(key="c", layout="n", mean=k["mean_c"], std=k["std_c"])]
        spec = DatasetSpec(file, "solution", 10512, 60, 240, ch, "[u],[c]", [False, False], k["time"], int(k["time"]), has_pixel_mask=False)
        dflt = dict(max_num_time_steps=steps, time_step_size=2)
    elif "reaction_diffusion" in name:
        if "reaction_diffusion.AllenCahn" not in name:
            raise ValueError(f"Unknown dataset {name}")   # (the reference falls through to an UnboundLocalError here)
        # reaction_diffusion/allen_cahn.py:6-53
        spec = DatasetSpec("/ACE.nc", "solution", 15000, 60, 240, [Channel(key="solution", layout="nt", mean=0.002484262, std=0.65351176)],
                           "[u]", [False], 19.0, 19, has_pixel_mask=False)
        dflt = dict(max_num_time_steps=9 if out else 7, time_step_size=2)
    else:
        raise ValueError(f"Unknown dataset {name}")
    return spec, dflt


def _kolmogorov(just_velocities: bool) -> DatasetSpec:
    """fluids/incompressible.py:149-243: forced NS — velocity from `solution`, density 1 / pressure 0 planes, and the (fixed,
    normalised) forcing 0.1 sin(2 pi (x + y)) on linspace(0, 1, 128)^2 as an extra channel of inputs AND labels."""
    m, s = list(_NS["mean"]), list(_NS["std"])
    m[1], m[2], s[1], s[2] = -2.2424793e-13, 4.1510376e-12, 0.22017328, 0.22078253
    R = 128
    xs = np.linspace(0.0, 1.0, R, dtype=np.float32)          # (torch.linspace(0, 1, R) in fp32)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    f = (np.float32(0.1) * np.sin(np.float32(2.0 * np.pi) * (X + Y))).astype(np.float32)
    forcing = ((f - np.float32(-1.2996679288335145e-09)) / np.float32(0.0707106739282608)).astype(np.float32)
    vel = [Channel(src=0, key="solution", mean=m[1], std=s[1]), Channel(src=1, key="solution", mean=m[2], std=s[2])]
    if just_velocities:
        ch, desc, mask = vel + [Channel(plane=forcing)], "[u,v],[g]", [False, False, False]
    else:
        ch = [Channel(const=1.0, mean=m[0], std=s[0])] + vel + [Channel(const=0.0, mean=m[3], std=s[3]), Channel(plane=forcing)]
        desc, mask = "[rho],[u,v],[p],[g]", [False, False, False, True, False]
    return DatasetSpec("/FNS-KF.nc", "solution", 20000, 120, 240, ch, desc, mask, _NS["time"], 20)


def _airfoil() -> DatasetSpec:
    """fluids/compressible.py:8-53: steady flow around an airfoil — input = plane 0 of `solution` (1 inside the body), label =
    plane 1 normalised; pixel_mask = (input == 1) element-wise and the label is set to 1 there."""
    return DatasetSpec("/SE-AF.nc", "solution", 10869, 120, 240, [Channel(src=0, key="solution", layout="nk")], "[rho]", [False], 1.0, 0,
                       labels=[Channel(src=1, key="solution", layout="nk", mean=0.92984116, std=0.10864315)], steady=True,
                       has_pixel_mask=False, input_mask_value=1.0)


class NpyDirectory:
    """A dataset file exported as one `.npy` per array (`<dir>/<key>.npy`): every array is memory-mapped on first use, so a sample
    read touches only the pages of its own planes — the numpy-only twin of an HDF5 file for hosts without h5py."""

    def __init__(self, directory: str):
        self.directory = directory
        self._maps: Dict[str, np.ndarray] = {}

    def keys(self):
        return sorted(f[:-4] for f in os.listdir(self.directory) if f.endswith(".npy"))

    def __contains__(self, key):
        return os.path.exists(os.path.join(self.directory, key + ".npy"))

    def __getitem__(self, key):
        if key not in self._maps:
            path = os.path.join(self.directory, key + ".npy")
            if not os.path.exists(path):
                raise KeyError(f"{key} (no {path})")
            self._maps[key] = np.load(path, mmap_mode="r")
        return self._maps[key]


def export_npy(reader, keys: Sequence[str], directory: str) -> None:
    """write arrays `keys` of an open reader (h5py file, dict, …) as `<directory>/<key>.npy` for NpyDirectory"""
    os.makedirs(directory, exist_ok=True)
    for k in keys:
        np.save(os.path.join(directory, k + ".npy"), np.asarray(reader[k]))


def open_reader(path: str):
    """`.nc` / `.h5` files need h5py (HDF5 / netCDF-4).  numpy-only alternatives: `<path>` itself when it is a `.npy` (one
    memory-mapped array answering every key), a `.npz`, or a directory of per-array `.npy` files (NpyDirectory); and for a `.nc` /
    `.h5` name an export of it lying beside it (`<stem>/`, `<stem>.npy`, `<stem>.npz`) — used when the file itself is absent, and
    BEFORE the file on a host without h5py (the case the export exists for)."""
    stem = os.path.splitext(path)[0]
    direct = path.endswith((".npy", ".npz")) or os.path.isdir(path)
    have_h5py = True
    try:
        import h5py
    except ImportError:   # pragma: no cover - environment dependent
        have_h5py = False
    # with h5py the file the reference opens wins; without it (or when the file is absent) its numpy export lying beside it does
    order = (path,) if direct else ((path, stem, stem + ".npy", stem + ".npz") if have_h5py else (stem, stem + ".npy", stem + ".npz", path))
    for cand in order:
        if os.path.isdir(cand):
            return NpyDirectory(cand)
        if cand.endswith(".npy") and os.path.exists(cand):
            r = np.load(cand, mmap_mode="r")
            return {"data": r, "solution": r, "velocity": r}
        if cand.endswith(".npz") and os.path.exists(cand):
            return np.load(cand)
        if cand == path and os.path.exists(cand):
            break
    try:
        import h5py
    except ImportError as e:   # pragma: no cover - environment dependent
        raise ImportError(f"reading {path} needs h5py (not installed here); pass reader={{key: array}}, or export the arrays with "
                          f"poseidon_amd.data.export_npy to {stem}/ (one .npy per array), {stem}.npy or {stem}.npz") from e
    return h5py.File(path, "r")


class _SampleGroups:
    """The Helmholtz file keeps one group per sample ("Sample_<i>" with datasets a, bc, u: elliptic/helmholtz.py:33-45); this view
    indexes it like the other readers: view[key][i]."""

    class _Key:
        def __init__(self, reader, key):
            self.reader, self.key = reader, key

        def __getitem__(self, i):
            return np.asarray(self.reader["Sample_" + str(int(i))][self.key])

    def __init__(self, reader):
        self.reader = reader

    def __getitem__(self, key):
        return _SampleGroups._Key(self.reader, key)


def _downsample(image: torch.Tensor, target: int) -> torch.Tensor:
    """the readers' own spectral downsampling (fluids/incompressible.py:75-83; the twin of ScOT._downsample), CPU path"""
    image = image.unsqueeze(0)
    n = image.shape[-2]
    freqs = torch.fft.fftfreq(n, d=1 / n)
    sel = torch.logical_and(freqs >= -target / 2, freqs <= target / 2 - 1)
    hat = torch.fft.fft2(image, norm="forward")[:, :, sel, :][:, :, :, sel]
    return torch.fft.ifft2(hat, norm="forward").real.squeeze(0)


# ------------------------------------------------------------------------------------------------ datasets
class PDEDataset(torch.utils.data.Dataset):
    """One dataset of the reference's registry with the reference's sample dict; `to_device()` gives the HBM-resident twin."""

    def __init__(self, name: str, which: str = "train", num_trajectories: int = -1, data_path: str = "./data", reader=None,
                 max_num_time_steps: Optional[int] = None, time_step_size: Optional[int] = None,
                 fix_input_to_time_step: Optional[int] = None, allowed_time_transitions: Optional[Sequence[int]] = None,
                 n_max: Optional[int] = None, n_val: Optional[int] = None, n_test: Optional[int] = None,
                 just_velocities: bool = False, resolution: Optional[int] = None, **_):
        spec, dflt = _spec(name, just_velocities)
        if just_velocities:
            if "fluids.incompressible" not in name:
                raise TypeError("just_velocities is an option of the incompressible readers")
            if "KolmogorovFlow" not in name:     # fluids/incompressible.py:44-63: drop the constant density / pressure planes
                keep = [k for k, c in enumerate(spec.channels) if c.is_array()]
                spec.channels = [spec.channels[k] for k in keep]
                spec.pixel_mask = [False] * len(keep)
                spec.label_description = "[u,v]" + (",[tracer]" if "tracer" in name else "")
        if resolution is not None:
            if "fluids.incompressible" not in name or "KolmogorovFlow" in name:
                raise TypeError("resolution is an option of the incompressible readers")
            if resolution > 128:
                raise ValueError("Resolution must be <= 128")
        self.res = resolution
        self.name, self.spec, self.which = name, spec, which
        self.time_wrapped = spec.steady and ".time" in name          # base.py:372-395: time = 1.0
        self.steady = spec.steady
        if spec.steady:
            self.pairs = TimePairs(1, 1)
            self.pairs.multiplier = 1
        else:
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
        self.label_channels = spec.labels if spec.labels is not None else spec.channels
        self.label_description = spec.label_description
        self.output_dim = spec.label_description.count(",") + 1
        self.printable_channel_description, self.channel_slice_list = channel_lists(spec.label_description)
        self.pixel_mask = torch.tensor(spec.pixel_mask) if spec.has_pixel_mask else None
        self.reader = reader if reader is not None else open_reader(os.path.join(data_path, spec.file.lstrip("/")))
        if "elliptic.Helmholtz" in name and "a" not in getattr(self.reader, "keys", lambda: [])():
            self.reader = _SampleGroups(self.reader)

    def __len__(self) -> int:
        return self.length

    def _plane(self, c: Channel, i: int, t: Optional[int]) -> torch.Tensor:
        R = self.resolution
        if c.plane is not None:
            return torch.from_numpy(np.asarray(c.plane, dtype=np.float32))
        if not c.is_array():
            return (torch.full((R, R), float(c.const), dtype=torch.float32) - c.mean) / c.std
        arr = self.reader[c.key or self.spec.key]
        j = i + self.start
        if c.layout == "ntc":
            x = arr[j, t, c.src:c.src + 1]
        elif c.layout == "nt":
            x = arr[j, t]
        elif c.layout == "nk":
            x = arr[j, c.src]
        elif c.layout == "n":
            x = arr[j]
        else:   # "s": one number per sample
            x = np.full((R, R), float(np.asarray(arr[j])), dtype=np.float32)
        x = torch.from_numpy(np.array(x, dtype=np.float32)).reshape(R, R)      # (a copy: memory-mapped sources are read-only)
        if self.spec.transpose:
            x = x.transpose(-2, -1)
        if c.shift:
            x = x - c.shift
        return (x - c.mean) / c.std

    def _planes(self, channels, i: int, t: Optional[int]) -> torch.Tensor:
        return torch.stack([self._plane(c, i, t) for c in channels], 0)

    def __getitem__(self, idx: int) -> Dict:
        if self.steady:
            pv, lab = self._planes(self.spec.channels, idx, None), self._planes(self.label_channels, idx, None)
            out = {"pixel_values": pv, "labels": lab}
            if self.spec.input_mask_value is not None:      # airfoil: element-wise mask (1, H, W); the label is 1 inside the body
                mask = pv == self.spec.input_mask_value
                lab[mask] = self.spec.input_mask_value
                out["pixel_mask"] = mask
            if self.time_wrapped:
                out["time"] = 1.0
            return out
        i, t, t1, t2 = self.pairs(idx)
        pv, lab = self._planes(self.spec.channels, i, t1), self._planes(self.label_channels, i, t2)
        if self.res is not None:
            pv, lab = _downsample(pv, self.res), _downsample(lab, self.res)
        out = {"pixel_values": pv, "labels": lab, "time": t / self.spec.time_const}
        if self.pixel_mask is not None:
            out["pixel_mask"] = self.pixel_mask
        return out

    def to_device(self, device="cuda", trajectories: Optional[int] = None) -> "DeviceTrajectories":
        return DeviceTrajectories(self, device, trajectories)


class DeviceTrajectories:
    """This split's trajectories resident in HBM; `batch(indices)` -> the collated dict of a reference batch in one launch (two when
    inputs and labels follow different recipes).  Every array a recipe reads becomes a source plane of ONE resident tensor
    data[n, T, nsrc, H, W] (planes that are static in time are repeated along T; steady datasets have T = 1)."""

    def __init__(self, ds: PDEDataset, device="cuda", trajectories: Optional[int] = None):
        self.ds = ds
        spec = ds.spec
        n = ds.trajectories if trajectories is None else min(trajectories, ds.trajectories)
        R = ds.resolution
        main = ds.reader[spec.key]
        self.T = 1 if spec.steady else int(np.asarray(main[ds.start]).shape[0])
        sources, planes = [], []

        def source_of(c: Channel) -> int:
            if c.plane is not None:
                planes.append(np.asarray(c.plane, dtype=np.float32))
                return -1 - len(planes)                   # -2 - p: fixed plane p
            if not c.is_array():
                return -1
            k = (c.key or spec.key, c.layout, c.src if c.layout in ("ntc", "nk") else None)
            if k not in sources:
                sources.append(k)
            return sources.index(k)
        self.symmetric = spec.labels is None
        src_in = [source_of(c) for c in spec.channels]
        src_lab = src_in if self.symmetric else [source_of(c) for c in ds.label_channels]
        self.nsrc = max(1, len(sources))
        host = np.zeros((n, self.T, self.nsrc, R, R), dtype=np.float32)
        for q, (key, layout, src) in enumerate(sources):
            arr = ds.reader[key]
            if layout == "ntc":
                host[:, :, q] = np.asarray(arr[ds.start:ds.start + n, :, src], dtype=np.float32).reshape(n, self.T, R, R)
            elif layout == "nt":
                host[:, :, q] = np.asarray(arr[ds.start:ds.start + n], dtype=np.float32).reshape(n, self.T, R, R)
            else:
                for j in range(n):
                    x = arr[ds.start + j, src] if layout == "nk" else arr[ds.start + j]
                    host[j, :, q] = np.asarray(x, dtype=np.float32).reshape((1, 1) if layout == "s" else (R, R))
        self.data = torch.from_numpy(host).to(device)
        self.n = n
        self.planes = torch.from_numpy(np.stack(planes)).to(device) if planes else None

        def recipe(channels, srcs):
            ab = [c.affine() for c in channels]
            return (torch.tensor(srcs, dtype=torch.int32, device=device), torch.tensor([x[0] for x in ab], dtype=torch.float32, device=device),
                    torch.tensor([x[1] for x in ab], dtype=torch.float32, device=device))
        self.rin = recipe(spec.channels, src_in)
        self.rlab = self.rin if self.symmetric else recipe(ds.label_channels, src_lab)
        self.pixel_mask = ds.pixel_mask.to(device) if ds.pixel_mask is not None else None

    def __len__(self) -> int:
        return self.n * self.ds.pairs.multiplier

    def batch(self, indices) -> Dict[str, torch.Tensor]:
        from . import ops
        ds = self.ds
        idx = np.asarray(indices, dtype=np.int64)
        if ds.steady:
            i, t1, t2 = idx, np.zeros_like(idx), np.zeros_like(idx)
        else:
            i, t1, t2 = ds.pairs.arrays(idx)
        if idx.size == 0 or i.max() >= self.n or t2.max() >= self.T:
            raise IndexError("sample index outside the resident trajectories")
        dev = self.data.device
        it = torch.from_numpy(np.stack([i, t1, t2], 0).astype(np.int32)).to(dev)
        B, R = idx.size, ds.resolution
        pv = torch.empty(B, len(ds.spec.channels), R, R, dtype=torch.float32, device=dev)
        lab = torch.empty(B, len(ds.label_channels), R, R, dtype=torch.float32, device=dev)
        tr = bool(ds.spec.transpose)
        if self.symmetric and self.planes is None:
            ops.gather_pairs(self.data, it, *self.rin, pv, lab, self.T, self.nsrc, R, R, tr)
        else:
            ops.gather_planes(self.data, it[0], it[1], *self.rin, self.planes, pv, self.T, self.nsrc, R, R, tr)
            ops.gather_planes(self.data, it[0], it[2], *self.rlab, self.planes, lab, self.T, self.nsrc, R, R, tr)
        out = {"pixel_values": pv, "labels": lab}
        if ds.spec.input_mask_value is not None:
            mask = pv == ds.spec.input_mask_value
            lab.masked_fill_(mask, ds.spec.input_mask_value)
            out["pixel_mask"] = mask
        if not ds.steady:
            if ds.res is not None:
                from scOT.model import spectral_resize
                out["pixel_values"], out["labels"] = spectral_resize(pv, ds.res), spectral_resize(lab, ds.res)
            out["time"] = torch.from_numpy(((t2 - t1) / ds.spec.time_const).astype(np.float32)).to(dev)
        elif ds.time_wrapped:
            out["time"] = torch.ones(B, dtype=torch.float32, device=dev)
        if self.pixel_mask is not None:
            out["pixel_mask"] = self.pixel_mask.unsqueeze(0).expand(B, -1)
        return out


class _TimeDependent(type(PDEDataset)):
    def __instancecheck__(cls, obj):
        return isinstance(obj, PDEDataset) and (not obj.steady or obj.time_wrapped)


class BaseDataset(PDEDataset):
    """reference base.py:163-285 as a type: every dataset of the registry is one (`isinstance(ds, BaseDataset)`)"""


class BaseTimeDataset(PDEDataset, metaclass=_TimeDependent):
    """reference base.py:288-369 / 372-395 as a type: `isinstance(ds, BaseTimeDataset)` is how the reference's drivers ask whether
    samples carry a `time` (train.py:227-231, inference.py:74,224) — true for the time-dependent readers and for `.time`-wrapped ones."""


def get_dataset(dataset, **kwargs):
    """reference `get_dataset` (base.py:15-160): name -> dataset (".out": more time steps, ".tracer", ".time": a time-independent
    dataset with time = 1.0); a list of names -> torch ConcatDataset."""
    if isinstance(dataset, (list, tuple)):
        return torch.utils.data.ConcatDataset([get_dataset(d, **kwargs) for d in dataset])
    return BaseDataset(dataset, **kwargs)
