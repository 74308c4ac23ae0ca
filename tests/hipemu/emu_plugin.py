"""pytest plugin: run the GPU kernel parity tests (tests/test_kernels_gpu.py, unmodified, full shapes) on the CPU emulation.

    PYTHONPATH=tests/hipemu python -m pytest tests/test_kernels_gpu.py -p emu_plugin -q        # ~10 min, 138 passed at round 1

Process-wide patching (unlike the per-test monkeypatch of tests/test_kernels_emu_cpu.py): only for this manual run."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def pytest_configure(config):
    import emu_session
    from poseidon_amd import ops
    lib = emu_session.load_emu()
    ws = torch.empty(96 << 20, dtype=torch.uint8)
    ops.L = lambda: lib
    ops.ptr = lambda t: None if t is None else t.data_ptr()
    ops.stream = lambda: None
    ops.workspace = lambda need=0: ws
    torch.cuda.synchronize = lambda *a, **k: None
    os.environ["SCOT_EXPERIMENTAL"] = "1"


def pytest_collection_modifyitems(config, items):
    import test_kernels_gpu as G
    G.DEV = "cpu"
