"""Build libscot_hip.so (hipcc, gfx950) in-tree.  `python -m poseidon_amd.build [--force]`."""
from __future__ import annotations

import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["gemm.hip", "gemm_fast.hip", "gemm_wide.hip", "wgrad_wide.hip", "gemm_panel.hip", "attention.hip", "attention_w16.hip", "norm.hip", "norm_fast.hip", "mlp_fused.hip", "wgrad_mlp.hip", "host_tape.hip", "misc.hip", "optim.hip", "dp.hip"]
LIB = os.path.join(HERE, "libscot_hip.so")
# the same sources with the 16-bit operand type meaning IEEE binary16 instead of bfloat16 (csrc/common.h)
LIB_F16 = os.path.join(HERE, "libscot_hip_f16.so")
VARIANTS = [(LIB, "", []), (LIB_F16, ".f16", ["-DSCOT_OPERAND_FP16"])]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]


def _stale() -> bool:
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    for lib, _, _ in VARIANTS:
        if not os.path.exists(lib):
            return True
        t = os.path.getmtime(lib)
        if any(os.path.getmtime(d) > t for d in deps):
            return True
    return False


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every kernel source for gfx950, once per operand format, and link the two shared libraries."""
    if not force and not _stale():
        return LIB
    t0 = time.time()
    procs, objs = [], {lib: [] for lib, _, _ in VARIANTS}
    for lib, suffix, defs in VARIANTS:
        for src in SOURCES:
            obj = os.path.join(CSRC, src.replace(".hip", suffix + ".o"))
            objs[lib].append(obj)
            cmd = [HIPCC, *FLAGS, *defs, "-c", os.path.join(CSRC, src), "-o", obj]
            procs.append((src + suffix, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    for lib, _, _ in VARIANTS:
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs[lib], "-o", lib])
    if verbose:
        print(f"[poseidon_amd.build] built {LIB} and {LIB_F16} in {time.time() - t0:.0f}s", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
