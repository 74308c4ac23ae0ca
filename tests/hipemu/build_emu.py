"""Build libscot_emu.so: every kernel source of poseidon_amd/csrc compiled as HOST C++ against the hipemu shim (test
infrastructure; see hip/hip_runtime.h).  The sources are copied to a scratch directory with ONE textual rewrite —
`extern __shared__ T name[];` (dynamic LDS) becomes a pointer to the shim's buffer — and compiled with the ROCm clang."""
import glob
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(os.path.dirname(HERE)), "poseidon_amd", "csrc")
CLANG = os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
DYN = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?(\w+)\s+(\w+)\[\];")


def build(out_dir: str, sanitize: str = "", opt: str = "-O1", defines=()) -> str:
    src_dir = os.path.join(out_dir, "src")
    os.makedirs(src_dir, exist_ok=True)
    units = []
    for path in sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip"))):
        text = open(path).read()
        text = DYN.sub(lambda m: f"{m.group(1)}* {m.group(2)} = ({m.group(1)}*)hipemu::dyn_smem();", text)
        dst = os.path.join(src_dir, os.path.basename(path))
        with open(dst, "w") as f:
            f.write(text)
        if dst.endswith(".hip"):
            units.append(dst)
    flags = [CLANG, "-x", "c++", "-std=c++20", opt, "-g", "-fPIC", "-pthread", "-I", HERE, "-I", src_dir, "-Wno-unused-value",
             "-Wno-unknown-attributes"] + [f"-D{d}" for d in defines] + ([f"-fsanitize={sanitize}"] if sanitize else [])
    if sanitize or os.environ.get("HIPEMU_THREADS") == "1":
        flags.append("-DHIPEMU_THREADS")     # work-items as OS threads: what the sanitizers can reason about

    def cc(u):
        obj = u[:-4] + ".o"
        r = subprocess.run(flags + ["-c", u, "-o", obj], capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"{os.path.basename(u)}:\n{r.stderr[-4000:]}")
        return obj
    with ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(cc, units))
    lib = os.path.join(out_dir, "libscot_emu.so")
    link = [CLANG, "-shared", "-pthread", "-o", lib] + objs + ([f"-fsanitize={sanitize}"] if sanitize else [])
    r = subprocess.run(link, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr[-4000:])
    return lib


def build_cached(sanitize: str = "", kind: str = "bf16") -> str:
    """Build under tests/hipemu/_build/<hash of sources + shim + flags> (git-ignored) and reuse it while nothing changed.
    kind = "f16": the -DSCOT_OPERAND_FP16 build (libscot_hip_f16.so's twin)."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")) +
                       [os.path.join(HERE, "hip", "hip_runtime.h"), os.path.abspath(__file__)]):
        h.update(open(path, "rb").read())
    h.update(sanitize.encode())
    gen = os.path.join(HERE, "_build", h.hexdigest()[:16])
    out = os.path.join(gen, kind)
    lib = os.path.join(out, "libscot_emu.so")
    if os.path.exists(lib):
        return lib
    if not os.path.isdir(gen):
        import shutil
        shutil.rmtree(os.path.join(HERE, "_build"), ignore_errors=True)   # one generation only
    return build(out, sanitize=sanitize, defines=("SCOT_OPERAND_FP16",) if kind == "f16" else ())


if __name__ == "__main__":
    print(build(sys.argv[1] if len(sys.argv) > 1 else "/tmp/hipemu_build", sanitize=sys.argv[2] if len(sys.argv) > 2 else ""))
