"""bench.py's main() end to end on CPU: world size 2 over gloo, the CPU-emulated kernels (tests/hipemu) and a tiny model — the launcher
contract (RANK / WORLD_SIZE / MASTER_* from the environment, --gpus must equal the group size), the rank-0-only sections, the `--dp auto`
probes with their timing all-reduces, `grad_exchange_exposed`, the communication timing, and the ONE JSON line.  A rank that skips a
collective another rank enters (the classic rank-0-only bug) deadlocks here, under the timeout, instead of on the 8-GPU box."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run(world, extra, tmp_path, timeout=900):
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    import build_emu
    if not (os.path.exists(build_emu.CLANG) or shutil.which(build_emu.CLANG)):
        pytest.skip("no host clang with __bf16 vector support")
    build_emu.build_cached()                  # once, before the ranks race for it
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SCOT_SIDE_STREAM="0", OMP_NUM_THREADS="2")
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--_emu", "--gpus", str(world), "--model", "T", "--size", "32", "--batch", "2",
               "--compute", "fp32", "--steps", "2", "--warmup", "1"] + extra
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=str(tmp_path)))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        pytest.fail(f"bench.py deadlocked with world size {world} ({extra})")
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}: rc {p.returncode}\n{se[-3000:]}"
    lines0 = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines0) == 1, outs[0][0][-2000:]                       # rank 0 prints exactly one JSON line
    for so, _ in outs[1:]:
        assert not [l for l in so.splitlines() if l.startswith("{")]   # ... and nobody else prints one
    return json.loads(lines0[0])


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("dp", ["auto", "overlap"])      # ("after" is the other arm of the auto probe)
def test_two_rank_bench_flow(tmp_path, dp):
    d = _run(2, ["--dp", dp], tmp_path)
    assert d["emulated"] and d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and abs(d["value"] - 4 / (d["ms_per_step"] / 1e3)) < 1e-6 * d["value"]
    assert d["config"]["grad_exchange"] in ("overlap", "after") and (dp == "auto" or d["config"]["grad_exchange"] == dp)
    if dp == "auto":
        assert "probe_dp_after_ms" in d["config"] and "probe_dp_overlap_ms" in d["config"]
    assert d["grad_exchange_exposed"] is not None and "step_without_exchange_ms" in d["grad_exchange_exposed"]
    assert d["grad_comm_ms_per_step"] is not None
    assert d["roofline"]["bound"] == "mfma" and "cpu_baseline" not in d      # (cpu_baseline is a world-size-1 section)


@pytest.mark.timeout(900)
def test_one_rank_group_and_bare_flow(tmp_path):
    """torchrun with one rank (the collective path with a group of one) and the bare single-process call"""
    d = _run(1, ["--dp", "auto"], tmp_path)
    assert d["n_gpus"] == 1 and d["config"]["grad_exchange"] in ("overlap", "after")


def test_group_size_must_match_gpus(tmp_path):
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--_emu", "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
