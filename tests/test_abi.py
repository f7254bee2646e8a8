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


def lib_rows(code):
    from genomad_amd import _lib
    return _lib.load().gnn_fused_rows_per_step(code)


def test_precision_enum_matches_the_binding_and_rows_per_step():
    """include/genomad_nn.h's gnn_precision values are what genomad_amd/_lib.py sends, and the (GPU-free) geometry query
    answers for every mode: 128 rows per fused step (32 * GNN_C6_NMB for f16c6, 96 = 32 Toom-Cook tiles for f16x3tc), 0 for the
    unfused f32 path."""
    from genomad_amd import _lib
    text = open(os.path.join(ROOT, "include", "genomad_nn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    enum = {name.lower(): int(val) for name, val in re.findall(r"GNN_PREC_([A-Z0-9]+)\s*=\s*(\d+)", text)}
    # the two arithmetics round 6 removed keep their enum values (old callers get GNN_ERR_STATE) and have no name on the Python side
    assert enum.pop("bf16") == _lib.PREC_BF16 == 2 and enum.pop("f16c8") == _lib.PREC_F16C8 == 3
    assert enum == _lib.PRECISIONS
    assert lib_rows(_lib.PREC_BF16) < 0 and lib_rows(_lib.PREC_F16C8) < 0
    lib = _lib.load()
    for name, val in _lib.PRECISIONS.items():
        rows = lib.gnn_fused_rows_per_step(val)
        assert rows == (0 if name == "f32" else 96 if name in ("f16x3tc", "f16x3tk") else 128 if name != "f16c6" else rows) and rows % 32 == 0
    assert lib.gnn_fused_rows_per_step(_lib.PRECISIONS["f16c6"]) in (128, 160)
    assert lib.gnn_fused_rows_per_step(77) < 0
    from genomad_amd import nn_classification
    assert nn_classification.DEFAULT_PRECISION in _lib.PRECISIONS


def _e2m3_value(code):
    import numpy as np
    code = np.asarray(code)
    e, m = (code >> 3) & 3, code & 7
    mag = np.where(e == 0, m / 8.0, (1 + m / 8.0) * np.exp2(e - 1.0))
    return np.where(code & 32, -mag, mag)


def test_f16c6_weight_stream_decodes_to_the_mx_rounding_of_the_oracle_study():
    """gnn_debug_pack_c6 (the host packer gnn_load_weights uses for GNN_PREC_F16C6) against an independent numpy
    decoding: the f16 fragments are f16(w) (RNE) in MFMA fragment order, the fp6 fragments times 2^(scale byte - 127)
    are the OCP-MX e2m3 images of w and of w - f16(w) (block = the 32 k of a step, exponent floor(log2 amax) - 2) that
    oracle/precision_study.py's rnd_mx emulates — the arithmetic DESIGN.md section 2 prices."""
    import numpy as np
    from genomad_amd import _lib
    from oracle import precision_study as PS
    lib = _lib.load()
    rng = np.random.default_rng(5)
    K, N = 256, 64
    w = (rng.standard_normal((K, N)) * np.exp2(rng.integers(-6, 3, size=(K, 1)))).astype(np.float32)
    w[7, 3] = 0.0
    w[32:64, 5] = 0.0                                         # an all-zero MX block
    need = ctypes.c_size_t()
    fp = w.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    _lib.check(lib.gnn_debug_pack_c6(fp, K, N, None, 0, ctypes.byref(need)))
    nk32, nblk = K // 32, N // 32
    assert need.value == nk32 * nblk * 3584 // 4 + (K // 128) * nblk * 64
    out = np.zeros(need.value, np.uint32)
    _lib.check(lib.gnn_debug_pack_c6(fp, K, N, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), out.size, None))
    frag = out[:nk32 * nblk * 896].view(np.uint8).reshape(nk32, nblk, 3584)
    scales = out[nk32 * nblk * 896:].view(np.uint8).reshape(K // 128, nblk, 64, 4)
    h16 = w.astype(np.float16)
    want_w = PS.rnd_mx(w.astype(np.float64), "e2m3", axis=0)
    want_lo = PS.rnd_mx(w.astype(np.float64) - h16.astype(np.float64), "e2m3", axis=0)
    for ks in range(nk32):
        for nb in range(nblk):
            rec = frag[ks, nb]
            for s in range(2):                                 # f16 fragments
                got = rec[s * 1024:(s + 1) * 1024].view(np.float16).reshape(64, 8)
                for lane in (0, 13, 31, 32, 47, 63):
                    k0 = ks * 32 + s * 16 + (lane >> 5) * 8
                    assert np.array_equal(got[lane], h16[k0:k0 + 8, nb * 32 + (lane & 31)])
            a = rec[2048:3072].reshape(64, 16)
            b = rec[3072:3584].reshape(64, 8)
            bits = np.unpackbits(np.concatenate([a, b], axis=1), axis=1, bitorder="little").reshape(64, 32, 6)
            codes = (bits * (1 << np.arange(6))).sum(axis=2)
            val = _e2m3_value(codes) * np.exp2(scales[ks // 4, nb, :, ks % 4].astype(np.float64) - 127)[:, None]
            cols = nb * 32 + (np.arange(64) & 31)
            want = np.where((np.arange(64) >> 5)[:, None] == 0, want_lo[ks * 32:(ks + 1) * 32, cols].T,
                            want_w[ks * 32:(ks + 1) * 32, cols].T)
            assert np.array_equal(val, want), (ks, nb)
    with pytest.raises(_lib.GnnError):
        _lib.check(lib.gnn_debug_pack_c6(fp, 100, N, None, 0, ctypes.byref(need)))
