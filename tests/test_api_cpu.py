"""CPU-only tests of the boundary: state-dict schema, config JSON, checkpoint round trip, C-ABI symbol export,
loud failure without a GPU."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_fixture
from poseidon_amd import lib as scotlib
from poseidon_amd.config import ScOTConfig, preset
from poseidon_amd.geometry import count_params, param_shapes, stage_plan
from poseidon_amd.synth import synth_state_dict
from scOT.model import ConditionalLayerNorm, LayerNorm, ScOT, ScOTOutput


def test_param_counts_match_survey():
    assert count_params(preset("T", image_size=128, num_channels=4, num_out_channels=4)) == 20774444
    assert count_params(preset("B", image_size=128, num_channels=4, num_out_channels=4)) == 157729988
    assert count_params(preset("L", image_size=128, num_channels=4, num_out_channels=4)) == 628575524


def test_state_dict_schema_matches_reference_fixture():
    f, meta = load_fixture("tiny_trained")
    cfg = ScOTConfig(**meta["cfg"])
    model = ScOT(cfg)
    ref_keys = [k[5:] for k in f.files if k.startswith("grad:")]
    assert list(model.state_dict().keys()) == ref_keys
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(f["grad:" + k].shape)
    f2, meta2 = load_fixture("tiny_nocond_p2")
    m2 = ScOT(ScOTConfig(**meta2["cfg"]))
    assert list(m2.state_dict().keys()) == [k[5:] for k in f2.files if k.startswith("grad:")]
    assert any(isinstance(m, LayerNorm) for m in m2.modules()) and not any(isinstance(m, ConditionalLayerNorm) for m in m2.modules())


def test_geometry_poseidon_b():
    cfg = preset("B", image_size=128, num_channels=4, num_out_channels=4)
    grid, enc, dec = stage_plan(cfg)
    assert grid == (32, 32)
    ws = [[b.window_shift() for b in st.blocks] for st in enc]
    assert ws[0][:4] == [(16, 0), (16, 8), (16, 0), (16, 8)] and ws[1][1] == (16, 0) and ws[2][1] == (8, 0) and ws[3][1] == (4, 0)
    assert [b.window_shift() for b in dec[3].blocks][:3] == [(16, 8), (16, 0), (16, 8)]  # reversed construction
    cfg5 = preset("B", image_size=256, num_channels=4, num_out_channels=4)
    _, enc5, _ = stage_plan(cfg5)
    assert [st.blocks[1].window_shift() for st in enc5] == [(16, 8), (16, 8), (16, 0), (8, 0)]


def test_config_json_roundtrip(tmp_path):
    cfg = preset("T", image_size=128, num_channels=4, num_out_channels=4, channel_slice_list_normalized_loss=[0, 1, 3, 4])
    cfg.save_pretrained(str(tmp_path))
    d = json.load(open(tmp_path / "config.json"))
    assert d["model_type"] == "swinv2" and d["hidden_size"] == 384 and d["num_layers"] == 4
    cfg2 = ScOTConfig.from_pretrained(str(tmp_path))
    assert cfg2.to_dict() == cfg.to_dict()
    assert ScOTConfig(learn_residual=True, use_conditioning=False).learn_residual is False  # reference model.py:122


@pytest.mark.parametrize("safe", [True, False])
def test_checkpoint_roundtrip_and_mismatched_sizes(tmp_path, safe):
    f, meta = load_fixture("tiny_trained")
    cfg = ScOTConfig(**meta["cfg"])
    model = ScOT(cfg)
    model.load_state_dict(synth_state_dict(param_shapes(cfg), "trained"))
    model.save_pretrained(str(tmp_path), safe_serialization=safe)
    assert os.path.exists(tmp_path / ("model.safetensors" if safe else "pytorch_model.bin"))
    m2 = ScOT.from_pretrained(str(tmp_path))
    for (k, a), (_, b) in zip(model.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    cfg3 = ScOTConfig(**dict(meta["cfg"], num_channels=5, num_out_channels=3, channel_slice_list_normalized_loss=[0, 1, 3]))
    with pytest.raises(RuntimeError):
        ScOT.from_pretrained(str(tmp_path), config=cfg3)
    m3 = ScOT.from_pretrained(str(tmp_path), config=cfg3, ignore_mismatched_sizes=True)
    assert sorted(m3._load_report["mismatched"]) == sorted([
        "embeddings.patch_embeddings.projection.weight", "patch_recovery.projection.weight",
        "patch_recovery.projection.bias", "patch_recovery.mixup.weight"])


def test_forward_fails_loudly_without_gpu():
    f, meta = load_fixture("tiny_trained")
    model = ScOT(ScOTConfig(**meta["cfg"]))
    with pytest.raises(scotlib.ScotLibraryError):
        model(pixel_values=torch.zeros(1, 4, 32, 32), time=torch.zeros(1))
    with pytest.raises(ValueError):
        model(pixel_values=None)


def test_output_is_indexable_like_hf():
    o = ScOTOutput(loss=torch.tensor(1.0), output=torch.zeros(1))
    assert o["loss"] is o.loss and o[0] is o.loss and o[1] is o.output and "loss" in o


def test_c_abi_exports_every_declared_symbol():
    lib = scotlib.load()
    header = open(os.path.join(ROOT, "include", "scot_hip.h")).read()
    declared = set(re.findall(r"\b(scot_[a-z0-9_]+)\s*\(", header)) - {"scot_stream_t"}
    assert declared == set(scotlib.PROTOTYPES.keys())
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.scot_abi_version() == scotlib.ABI_VERSION and lib.scot_operand_format() == 0
    f16 = scotlib.load(kind="f16")          # the binary16 build of the same sources exports the same C ABI
    for name in declared:
        assert getattr(f16, name) is not None
    assert f16.scot_abi_version() == scotlib.ABI_VERSION and f16.scot_operand_format() == 1


def test_dp_entry_points_without_a_communicator():
    """`scot_dp_*` (csrc/dp.hip) outside init..finalize: world 0 / rank -1, the all-reduce refuses (-3: unsupported configuration),
    bad arguments are shape errors before RCCL is touched, finalize is idempotent — no GPU and no RCCL call involved."""
    import ctypes
    for kind in ("bf16", "f16"):
        lib = scotlib.load(kind=kind)
        assert lib.scot_dp_world() == 0 and lib.scot_dp_rank() == -1
        assert lib.scot_dp_allreduce_bucket(None, 16, 0, None) == -3
        buf = ctypes.create_string_buffer(128)
        assert lib.scot_dp_init(None, 0, 1) == -1 and lib.scot_dp_init(ctypes.cast(buf, ctypes.c_void_p), 2, 2) == -1
        assert lib.scot_dp_unique_id(None) == -1
        assert lib.scot_dp_finalize() == 0 and lib.scot_dp_finalize() == 0
    from poseidon_amd.dp import GradAllReducer
    with pytest.raises(ValueError):
        GradAllReducer(None, None, backend="native", collective="rs_ag")
    with pytest.raises(ValueError):
        GradAllReducer(None, None, backend="mpi")


def test_workspace_queries_answer_without_a_gpu():
    """SURVEY.md §8(b) `scot_<op>_workspace_bytes(dims…)`: the split policy of the launching entry points, asked without launching.
    Poseidon-B at batch 64: the stage-0 / stage-1 grouped weight gradients split K (partial tiles), the deep stages run unsplit."""
    import ctypes
    lib = scotlib.load()
    IA = ctypes.c_int * 4
    need = [lib.scot_wgrad_group_workspace_bytes(4, K, IA(C, 4 * C, C, 3 * C), IA(4 * C, C, C, C))
            for K, C in ((65536, 96), (16384, 192), (4096, 384), (1024, 768))]
    plane = [12 * C * C * 4 for C in (96, 192, 384, 768)]          # Σ M_i N_i floats of a layer's four weight gradients
    assert need[0] > 0 and need[0] % plane[0] == 0 and need[1] > 0 and need[1] % plane[1] == 0 and need[2:] == [0, 0]
    # round 5, csrc/wgrad_wide.hip (128 x 128 tiles): a group with < 256 tiles and >= 8192 tokens cuts K into slices — Poseidon-B at 256 x 256, stage 2
    q = lib.scot_wgrad_group_workspace_bytes(4, 8192, IA(384, 1536, 384, 1152), IA(1536, 384, 384, 384))
    assert q > 0 and q % plane[2] == 0
    lib.scot_gemm_wide_config(0, 0)          # the 96 x 96-tile grouped kernel splits this shape too, by its own policy (fewer slices)
    try:
        q0 = lib.scot_wgrad_group_workspace_bytes(4, 8192, IA(384, 1536, 384, 1152), IA(1536, 384, 384, 384))
        assert q0 % plane[2] == 0 and q0 != q
    finally:
        lib.scot_gemm_wide_config(1, 0)
    assert 8 <= need[0] // plane[0] <= 128                          # nsplit: enough K slices to fill the chip, >= 8 K-tiles each
    tn = lib.scot_gemm_workspace_bytes(2, 1, 96, 384, 65536)        # one long-K weight gradient alone (TN)
    assert tn > 0 and tn % (96 * 384 * 4) == 0
    assert lib.scot_gemm_workspace_bytes(0, 1, 1024, 768, 3072) == 0      # forward GEMMs run unsplit by default
    assert lib.scot_gemm_workspace_bytes(2, 1, 96, 384, 7) == 0            # a shape the fast path declines: no scratch either
    assert lib.scot_wgrad_group_workspace_bytes(0, 1024, IA(), IA()) == 0


def test_step_tape_skips_calls_the_library_declined():
    """A C-ABI call that answers SCOT_ERR_UNSUPPORTED launched nothing and its caller falls back to other launches: the recorded step
    must hold the fallback launches only (a replayed -3 would otherwise abort every later step)."""
    from poseidon_amd import ops

    class FakeLib:
        def scot_covered(self, *a):
            return 0

        def scot_declined(self, *a):
            return -3
    log = []
    rec = ops._Recording(FakeLib(), log)
    assert rec.scot_declined(1, 2) == -3 and rec.scot_covered(3) == 0
    assert [(f.__name__, a) for f, a in log] == [("scot_covered", (3,))]


def test_package_import_reserves_hardware_queues():
    """Importing the package (before the HIP runtime starts) asks for 8 hardware queues unless the user chose a number: with the
    default 4, an RCCL communicator's streams push the engine's two streams onto one queue (27.1 vs 21.0 ms/step measured)."""
    import subprocess
    import sys
    code = "import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); import poseidon_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.stdout.strip() == "8", out.stderr[-500:]
    code = "import os; os.environ['GPU_MAX_HW_QUEUES'] = '6'; import poseidon_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.stdout.strip() == "6"


def test_from_pretrained_resolves_hub_ids_through_the_local_cache(tmp_path, monkeypatch):
    """`ScOT.from_pretrained("camlab-ethz/Poseidon-B")` (reference train.py:331-333): a hub id goes through huggingface_hub — its local
    cache first; with no cache entry and no network the error names both options instead of computing anything."""
    import huggingface_hub
    from scOT.model import ScOT
    cfg = ScOTConfig(image_size=16, patch_size=4, num_channels=2, num_out_channels=2, embed_dim=8, depths=[1], num_heads=[1],
                     skip_connections=[0], window_size=4, mlp_ratio=2.0)
    ckpt = tmp_path / "snap"
    ScOT(cfg).save_pretrained(str(ckpt))
    seen = {}

    def fake_snapshot(repo_id, **kw):
        seen.update(repo_id=repo_id, **kw)
        return str(ckpt)
    monkeypatch.setattr(huggingface_hub, "snapshot_download", fake_snapshot)
    m = ScOT.from_pretrained("camlab-ethz/Poseidon-X")
    assert seen["repo_id"] == "camlab-ethz/Poseidon-X" and seen["local_files_only"] is True
    assert m.config.embed_dim == 8 and not m._load_report["missing"]

    def no_network(repo_id, **kw):
        raise OSError("offline")
    monkeypatch.setattr(huggingface_hub, "snapshot_download", no_network)
    with pytest.raises(FileNotFoundError, match="neither a checkpoint directory nor a hub repository"):
        ScOT.from_pretrained("camlab-ethz/Poseidon-X")
