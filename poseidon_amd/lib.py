"""ctypes binding of libscot_hip.so (the C ABI declared in include/scot_hip.h).

The product path has NO fallback: if the library is missing or a call returns a non-zero status a RuntimeError is
raised.  (`build.py` compiles it in-tree with hipcc for gfx950; the .so travels to the GPU box with the repo.)
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libscot_hip.so")
# The same sources built twice (build.py): the format of the 16-bit operand type is a compile-time property (csrc/common.h).
LIB_PATHS = {"bf16": LIB_PATH, "f16": os.path.join(_HERE, "libscot_hip_f16.so")}
OPERAND_FORMAT = {"bf16": 0, "f16": 1}
ABI_VERSION = 5      # scot_abi_version() of the library these prototypes describe (checked at load)

P, I, F, Z = c_void_p, c_int, c_float, c_size_t

# name -> argtypes (restype is int unless listed in _VOID)
PROTOTYPES = {
    "scot_abi_version": [],
    "scot_operand_format": [],
    "scot_scale_inplace": [P, Z, F, P, P],
    "scot_scale_inplace_dev": [P, Z, P, P, P],
    "scot_selftest_tr": [P],
    "scot_set_use_tr": [I],
    "scot_get_use_tr": [],
    "scot_gemm": [I, I, I, I, I, P, I, I, I, P, I, I, I, P, I, I, P, P, P, I, I, P, I, I, I, P, P, Z, I, P, P],
    "scot_wgrad_group": [I, I, I, P, P, P, P, P, P, P, Z, P, P, P],
    "scot_segments_scale": [P, P, I, P, P, P],
    "scot_gemm_workspace_bytes": [I, I, I, I, I],
    "scot_gemm_wide_config": [I, I],
    "scot_gemm_splitk_config": [I, I],
    "scot_wgrad_group_workspace_bytes": [I, I, P, P],
    "scot_window_attn_fwd": [I, P, P, P, P, P, I, I, I, I, I, I, I, P],
    "scot_window_attn_probs": [P, I, P, P, P, P, I, I, I, I, I, I, I, P],
    "scot_window_attn_bwd": [I, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P],
    "scot_window_attn_bwd_rep": [I, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, Z, Z, P],
    "scot_replica_reduce": [P, I, I, Z, P, I, I, P, P],
    "scot_cpb_fwd": [P, P, P, P, P, P, I, I, P],
    "scot_cpb_bwd": [P, P, P, P, P, P, P, P, P, I, I, P],
    "scot_cpb_fwd_batched": [P, P, I, I, P, P, P, P],
    "scot_cpb_bwd_batched": [P, P, I, I, I, I, P, P, P, P, P],
    "scot_cln_fwd": [P, I, P, I, P, I, P, I, P, P, P, P, P, P, P, I, I, I, F, P, P],
    "scot_mlp_block_fwd": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, F, P],
    "scot_mlp_block_bwd": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P],
    "scot_block_tail_bwd": [P] * 32 + [P, P, I, P, P, P, I, I, I, I, P],
    "scot_block_tail_workgroups": [I, I, I],
    "scot_partial_colsum": [P, I, I, P, P],
    "scot_partial_colsum_batch": [I, P, P, P, P, P],
    "scot_wgrad_mlp_workspace_bytes": [I, I, I],
    "scot_wgrad_mlp": [P, P, P, P, P, P, P, P, P, I, I, I, P, Z, I, P, P],
    "scot_transpose_cast": [P, P, P, I, I, P],
    "scot_block_tail_fwd": [P] * 33 + [I, P, I, I, I, I, F, P],
    "scot_memset_async": [P, I, Z, P],
    "scot_memcpy_async": [P, P, Z, P],
    "scot_event_record": [P, P],
    "scot_stream_wait_event": [P, P],
    "scot_tape_replay": [P, Z, P],
    "scot_proj_cln_fwd": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, F, P],
    "scot_proj_cln_bwd": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, P],
    "scot_cln_bwd": [P, I, P, I, P, P, P, P, P, P, I, P, P, P, P, P, I, I, I, P, Z, P, I, P],
    "scot_cln_bwd_workspace_bytes": [I, I, I, I],
    "scot_cln_bwd_finish": [P, I, I, I, P, P, P, P, P],
    "scot_add": [P, I, P, I, P, I, Z, Z, P],
    "scot_batch_sum": [P, I, P, I, Z, P],
    "scot_gather_pairs": [P, P, P, P, P, P, P, I, I, I, I, I, I, I, P],
    "scot_pow2_rescale": [P, I, P, P],
    "scot_colscale_dev": [P, P, P, P, I, I, I, P],
    "scot_axpy_dev": [P, P, Z, P, I, P],
    "scot_gather_planes": [P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P],
    "scot_mask_tokens": [P, P, P, I, I, P],
    "scot_mask_tokens_bwd": [P, P, P, I, I, P],
    "scot_copy2d": [P, I, P, I, I, I, I, I, I, I, P],
    "scot_space_to_depth": [P, P, I, P, I, I, I, I, I, I, P],
    "scot_depth_to_space": [P, I, P, I, I, I, I, I, I, I, I, P],
    "scot_patchify": [P, P, I, I, I, I, I, I, P],
    "scot_unpatchify": [P, I, P, P, I, I, I, I, I, I, I, P],
    "scot_nchw_channel_sum": [P, P, I, I, I, P],
    "scot_colsum": [P, I, P, I, P, I, I, I, P],
    "scot_scale_residual": [P, I, P, P, I, P, I, Z, I, P],
    "scot_dwconv7": [P, I, P, P, P, I, I, I, I, I, I, P],
    "scot_dwconv7_wgrad": [P, I, P, I, P, P, I, I, I, I, P],
    "scot_conv5": [P, P, P, I, I, I, I, I, P],
    "scot_conv5_wgrad": [P, P, P, I, I, I, I, P],
    "scot_head_finalize": [P, P, I, P, P, I, P, P, I, I, I, I, P],
    "scot_loss_finish": [P, P, I, I, P, P],
    "scot_loss_bwd": [P, P, P, I, P, P, P, I, I, P, P, I, I, I, I, P],
    "scot_spectral_apply": [P, P, P, P, I, I, I, P],
    "scot_dp_pack": [P, P, Z, F, P],
    "scot_dp_unpack": [P, P, Z, F, P],
    "scot_dp_unique_id": [P],
    "scot_dp_init": [P, I, I],
    "scot_dp_allreduce_bucket": [P, Z, I, P],
    "scot_dp_world": [],
    "scot_dp_rank": [],
    "scot_dp_finalize": [],
    "scot_optim_blocks": [Z],
    "scot_grad_sqnorm": [P, P, Z, P, P],
    "scot_clip_coef": [P, I, F, P, P],
    "scot_adamw_step": [P, P, P, P, P, Z, P, P, I, F, F, F, I, P, P, P, P],
    "scot_optim_finish": [P, P, P, F, F, I, F, P],
}
_VOID = {"scot_set_use_tr", "scot_gemm_wide_config", "scot_gemm_splitk_config"}
_SIZE = {"scot_gemm_workspace_bytes", "scot_wgrad_group_workspace_bytes", "scot_cln_bwd_workspace_bytes", "scot_wgrad_mlp_workspace_bytes"}      # return size_t


def restype(name):
    return None if name in _VOID else (c_size_t if name in _SIZE else c_int)

_libs = {}


class ScotLibraryError(RuntimeError):
    pass


def load(path: str = None, kind: str = "bf16"):
    """Load (once per build) and type the library.  Raises if it is absent — there is no CPU/PyTorch fallback."""
    if kind not in LIB_PATHS:
        raise ValueError(f"unknown library build {kind!r}")
    if kind in _libs:
        return _libs[kind]
    path = path or os.environ.get("SCOT_LIB_" + kind.upper()) or LIB_PATHS[kind]      # (SCOT_LIB_F16 / SCOT_LIB_BF16: tools/ablate_kernels.py)
    # torch bundles its own libamdhip64.so (dlopen'ed by path).  It MUST be resident before our library is loaded so
    # that our NEEDED libamdhip64.so.7 resolves (by SONAME) to the same runtime; loading ours first pulls in
    # /opt/rocm's copy as a SECOND HIP runtime whose streams/pointers are foreign to torch's.
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise ScotLibraryError(
            f"{path} not found: the scOT hot path is HIP-only. Build it with `python -m poseidon_amd.build` "
            "(hipcc --offload-arch=gfx950).")
    lib = ctypes.CDLL(path)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.argtypes = argtypes
        fn.restype = restype(name)
    if lib.scot_abi_version() != ABI_VERSION:
        raise ScotLibraryError(f"{path} implements ABI version {lib.scot_abi_version()}, these bindings expect {ABI_VERSION}; "
                               "rebuild with `python -m poseidon_amd.build --force`")
    if lib.scot_operand_format() != OPERAND_FORMAT[kind]:
        raise ScotLibraryError(f"{path} was built for operand format {lib.scot_operand_format()}, expected {OPERAND_FORMAT[kind]} "
                               f"({kind}); rebuild with `python -m poseidon_amd.build --force`")
    # tuning knob for A/B runs (tools/gpu_ab.sh): SCOT_GEMM_WIDE = "<mode>[,<kernel variant>]" -> scot_gemm_wide_config (include/scot_hip.h)
    w = os.environ.get("SCOT_GEMM_WIDE")
    if w:
        mode, _, split = w.partition(",")
        lib.scot_gemm_wide_config(int(mode), int(split or 0))
    k = os.environ.get("SCOT_GEMM_SPLITK")      # A/B runs: "<slices>[,<zeroed too>]" -> scot_gemm_splitk_config (-1 = never split)
    if k:
        sl, _, z = k.partition(",")
        lib.scot_gemm_splitk_config(int(sl), int(z or 1))
    _libs[kind] = lib
    return lib


_ERR = {-1: "bad shape", -2: "bad dtype", -3: "unsupported configuration", -4: "kernel launch failed"}


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise ScotLibraryError(f"{what} failed: {_ERR.get(rc, rc)}")
