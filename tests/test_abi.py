"""CPU checks of the C-ABI boundary: the library loads and exports every symbol that
include/genomad_nn.h declares (no compute calls, no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "genomad_nn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gnn_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for must in ("gnn_create", "gnn_load_weights", "gnn_tokenize", "gnn_onehot_dev", "gnn_classify",
                 "gnn_classify_dev", "gnn_segment_mean", "gnn_last_error", "gnn_destroy"):
        assert must in syms


def test_library_loads_and_exports_every_declared_symbol():
    from genomad_amd import _lib
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/genomad_nn.h but not exported"
    assert set(_lib.SIGNATURES) == set(declared_symbols())
    assert lib.gnn_version() >= 100


def test_struct_layouts_match_the_header():
    from genomad_amd import _lib
    p = ctypes.sizeof(ctypes.c_void_p)
    assert ctypes.sizeof(_lib.IglooWeights) == 6 * p
    assert ctypes.sizeof(_lib.DenseBN) == 6 * p
    assert ctypes.sizeof(_lib.Weights) == (6 + 12 + 12 + 2) * p
    assert ctypes.sizeof(_lib.Taps) == 10 * p


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from genomad_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_lib.GnnError, match="no CPU fallback"):
        _lib.load()


def test_weight_validation_rejects_bad_tensors(synth_weights):
    from genomad_amd import weights
    w = dict(synth_weights)
    weights.validate(w)
    bad = dict(w)
    bad["conv2_kernel"] = bad["conv2_kernel"][:, :, :64]
    with pytest.raises(ValueError, match="conv2_kernel"):
        weights.validate(bad)
    bad = dict(w)
    p = bad["iglooA_patches"].copy()
    p[0, 0, 0] = 5997
    bad["iglooA_patches"] = p
    with pytest.raises(ValueError, match="out of range"):
        weights.validate(bad)
    bad = dict(w)
    del bad["out_dense_bias"]
    with pytest.raises(ValueError, match="missing"):
        weights.validate(bad)


def test_precision_enum_matches_the_binding_and_rows_per_step():
    """include/genomad_nn.h's gnn_precision values are what genomad_amd/_lib.py sends, and the (GPU-free) geometry query
    answers for every mode: 128 rows per fused step (32 * GNN_C6_NMB for f16c6), 0 for the unfused f32 path."""
    from genomad_amd import _lib
    text = open(os.path.join(ROOT, "include", "genomad_nn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    enum = {name.lower(): int(val) for name, val in re.findall(r"GNN_PREC_([A-Z0-9]+)\s*=\s*(\d+)", text)}
    assert enum == _lib.PRECISIONS
    lib = _lib.load()
    for name, val in _lib.PRECISIONS.items():
        rows = lib.gnn_fused_rows_per_step(val)
        assert rows == (0 if name == "f32" else 128 if name != "f16c6" else rows) and rows % 32 == 0
    assert lib.gnn_fused_rows_per_step(_lib.PRECISIONS["f16c6"]) in (128, 160)
    assert lib.gnn_fused_rows_per_step(77) < 0
    from genomad_amd import nn_classification
    assert nn_classification.DEFAULT_PRECISION in _lib.PRECISIONS
