"""Host wrappers of the downstream score consumers.

``branch_attention`` and ``score_batch_correction`` mirror the reference functions of the same names
(genomad/modules/aggregated_classification.py:10-34, genomad/modules/score_calibration.py:15-43):
same arguments, float64 in and out; the arithmetic runs in libgenomad_nn_hip.so.
"""
import numpy as np

from ._lib import check


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


def branch_attention(engine, w, b1, b2, temperature: float = 2) -> np.ndarray:
    """aggregated_classification.branch_attention: w (n,), marker scores b1 (n,3), nn scores b2 (n,3)."""
    w = _f64(w).reshape(-1)
    n = len(w)
    b1, b2 = _f64(b1, (n, 3)), _f64(b2, (n, 3))
    out = np.empty((n, 3), dtype=np.float64)
    check(engine.lib.gnn_branch_attention(engine.ctx, w.ctypes.data, b1.ctypes.data, b2.ctypes.data, n,
                                          float(temperature), out.ctypes.data))
    return out


def specificity(x) -> float:
    """utils.specificity / utils.entropy (utils.py:328-357): 1 - H(x)/log2(n)."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    if not np.any(x) or n == 1:
        return 0.0
    p = x / x.sum()
    p = p[p != 0]
    return float((np.log2(n) - (-1 * np.dot(p, np.log2(p)))) / np.log2(n))


def score_batch_correction(engine, scores, composition, classifier, weights_file) -> np.ndarray:
    """score_calibration.score_batch_correction: the composition smoothing (:18-21) is three scalars on
    the host, the 6->20->20->3 tanh MLP and softmax run on the device."""
    composition = _f64(composition, (3,))
    smoothing = 1 - specificity(composition) * 0.3
    composition = composition * smoothing + (np.ones(3) / 3) * (1 - smoothing)
    if classifier not in {"marker", "aggregated", "nn"}:
        classifier = "aggregated"
    z = np.load(weights_file)
    k = [_f64(z[f"{name}_{i}_{classifier}"]) for i in (1, 2, 3) for name in ("kernel", "bias")]
    scores = _f64(scores)
    n = len(scores)
    scores = _f64(scores, (n, 3))
    out = np.empty((n, 3), dtype=np.float64)
    check(engine.lib.gnn_score_calibration(engine.ctx, scores.ctypes.data, composition.ctypes.data,
                                           k[0].ctypes.data, k[1].ctypes.data, k[2].ctypes.data,
                                           k[3].ctypes.data, k[4].ctypes.data, k[5].ctypes.data, n,
                                           out.ctypes.data))
    return out
