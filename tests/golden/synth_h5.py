"""Synthetic stand-ins for the PDEgym files (no reference code; shared by the pin generator and the tests).

`SynthFile(path)` answers the subset of the h5py.File interface the readers use — file[key][index...] and, for Helmholtz.h5,
file["Sample_<i>"][key] — with values that are a pure function of (file name, key, sample index): sample j of an array is
`default_rng([crc32(file:key), j]).standard_normal(per-sample shape)` in float32, so any number of trajectories (the reference's
N_max is up to 20000) is addressable without holding them.  Per-sample shapes follow the indices the readers apply
(scOT/problems/**: e.g. NS-*.nc velocity[i, t, 0:2] and [.., 2:3] -> (T, 3, 128, 128)).
"""
import fnmatch
import os
import zlib

import numpy as np

R = 128
SHAPES = [   # (file name pattern, {key: per-sample shape})
    ("NS-*.nc", {"velocity": (21, 3, R, R)}),
    ("FNS-KF.nc", {"solution": (21, 2, R, R)}),
    ("SE-AF.nc", {"solution": (2, R, R)}),
    ("CE-RM.nc", {"solution": (21, 4, R, R)}),
    ("GCE-RT.nc", {"solution": (11, 6, R, R)}),
    ("CE-*.nc", {"data": (21, 5, R, R)}),
    ("Wave-*.nc", {"solution": (21, R, R), "c": (R, R)}),
    ("ACE.nc", {"solution": (20, R, R)}),
    ("Poisson-Gauss.nc", {"source": (R, R), "solution": (R, R)}),
    ("Helmholtz.h5", {"a": (R, R), "bc": (), "u": (R, R)}),
]


def shapes_of(path):
    base = os.path.basename(path)
    for pat, keys in SHAPES:
        if fnmatch.fnmatch(base, pat):
            return base, keys
    raise KeyError(f"no synthetic layout for {base}")


class SynthArray:
    def __init__(self, base, key, shape):
        self.seed = zlib.crc32(f"{base}:{key}".encode())
        self.base, self.key, self.sample_shape = base, key, tuple(shape)
        self._last = (None, None)

    def sample(self, j):
        j = int(j)
        if self._last[0] != j:
            x = np.random.default_rng([self.seed, j]).standard_normal(self.sample_shape).astype(np.float32)
            if self.base == "SE-AF.nc":       # plane 0 of the airfoil file is the body indicator: exact ones inside a disc
                yy, xx = np.mgrid[0:R, 0:R]
                x[0][(yy - 64) ** 2 + (xx - 40 - j % 17) ** 2 < 400] = 1.0
            self._last = (j, x)
        return self._last[1]

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        first, rest = idx[0], idx[1:]
        if isinstance(first, slice):
            if first.stop is None:
                raise IndexError("open-ended trajectory slices are not addressable on a synthetic file")
            out = np.stack([self.sample(j) for j in range(first.start or 0, first.stop, first.step or 1)])
            return out[(slice(None),) + rest] if rest else out
        x = self.sample(first)      # (h5py hands out a fresh array per read: the readers modify what they get in place)
        return np.array(x[rest]) if rest else (x.copy() if x.ndim else x[()])

    def __array__(self, dtype=None, copy=None):
        raise TypeError("a synthetic array has no finite extent along the trajectory axis; index it")


class _Group(dict):
    pass


class SynthFile:
    def __init__(self, path, mode="r"):
        self.base, self.layout = shapes_of(path)
        self.arrays = {k: SynthArray(self.base, k, s) for k, s in self.layout.items()}

    def keys(self):
        return [] if self.base == "Helmholtz.h5" else list(self.arrays)

    def __contains__(self, k):
        return k in self.keys()

    def __getitem__(self, key):
        if self.base == "Helmholtz.h5":
            if not key.startswith("Sample_"):
                raise KeyError(key)
            j = int(key[len("Sample_"):])
            return _Group({k: np.array(a.sample(j)) for k, a in self.arrays.items()})
        return self.arrays[key]

    def close(self):
        pass


def summary(x):
    """what a pin keeps of one tensor: shape, three float64 moments (the weighted one is not invariant under transposes or channel
    permutations) and a coarse subsample"""
    a = np.asarray(x, dtype=np.float64)
    w = ((np.arange(a.shape[-2])[:, None] * 131 + np.arange(a.shape[-1])[None, :] * 7) % 17 - 8.0)
    cw = 1.0 + np.arange(a.shape[0])[:, None, None] if a.ndim == 3 else 1.0
    return {"shape": list(a.shape), "sum": float(a.sum()), "abs": float(np.abs(a).sum()), "wsum": float((a * w * cw).sum()),
            "sub": np.asarray(x, dtype=np.float32)[..., 5::50, 3::50].reshape(-1).tolist()}
