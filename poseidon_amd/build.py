"""Build libscot_hip.so (hipcc, gfx950) in-tree.  `python -m poseidon_amd.build [--force]`."""
from __future__ import annotations

import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["gemm.hip", "gemm_fast.hip", "gemm_panel.hip", "attention.hip", "attention_w16.hip", "norm.hip", "norm_fast.hip", "mlp_fused.hip", "misc.hip", "optim.hip"]
LIB = os.path.join(HERE, "libscot_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    objs = []
    t0 = time.time()
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [HIPCC, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    if verbose:
        print(f"[poseidon_amd.build] built {LIB} in {time.time() - t0:.0f}s", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
