"""Shared by the CPU emulation tests: build / load libscot_emu.so and point `poseidon_amd.ops` at it for one test."""
import ctypes
import os
import shutil

import pytest
import torch

import build_emu
from poseidon_amd import lib as scot_lib
from poseidon_amd import ops

_libs = {}


def load_emu(kind="bf16"):
    _lib = _libs.get(kind)
    if _lib is None:
        if not (os.path.exists(build_emu.CLANG) or shutil.which(build_emu.CLANG)):
            pytest.skip("no host clang with __bf16 vector support")
        lib = ctypes.CDLL(build_emu.build_cached(kind=kind))
        for name, argtypes in scot_lib.PROTOTYPES.items():
            fn = getattr(lib, name)          # every symbol of the C ABI must exist in the emulated build too
            fn.argtypes = argtypes
            fn.restype = scot_lib.restype(name)
        # the emulated transposing LDS read must satisfy the kernels' own self test (the contract validated on the GPU)
        assert lib.scot_selftest_tr(None) >= 0 and lib.scot_get_use_tr() == 1
        assert lib.scot_operand_format() == scot_lib.OPERAND_FORMAT[kind]
        _libs[kind] = _lib = lib
    return _lib


def patch_ops(monkeypatch, lib, workspace_bytes=32 << 20):
    """CPU tensors go down the same wrappers for the duration of one test (the product's `ops.ptr` refuses them).
    `lib` is the bf16 build; the fp16 build is loaded on demand when an engine selects it (`ops.use("f16")`)."""
    ws = torch.empty(workspace_bytes, dtype=torch.uint8)

    def L():
        l = lib if ops._active == "bf16" else load_emu(ops._active)
        return ops._Recording(l, ops._recorder) if ops._recorder is not None else l   # step tape
    monkeypatch.setattr(ops, "L", L)
    monkeypatch.setattr(ops, "ptr", lambda t: None if t is None else t.data_ptr())
    monkeypatch.setattr(ops, "stream", lambda: None)
    monkeypatch.setattr(ops, "workspace", lambda need=0: ws)
