"""Which intermediate carries the f16c6 / f16c8 score error: for the worst windows of a sample, the back end (logits GEMM,
softmax, attention sum, dense head) is re-run in float64 numpy on MIXED device intermediates — m (IGLOO pair products) from one
path, yp (pooled y @ w_v) from the other.  Usage: error_paths.py [n_windows] [n_worst]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from genomad_amd import synthetic
from genomad_amd.engine import NNEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = int(sys.argv[2]) if len(sys.argv) > 2 else 24
W = synthetic.synth_weights()
w = {a: np.asarray(b, np.float64) if np.asarray(b).dtype.kind == "f" else np.asarray(b) for a, b in W.items()}
eng = NNEngine(0, W, chunk=512)

def softmax(x):
    e = np.exp(x - x.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)

def bn(x, g, b, mu, var):
    return g * (x - mu) / np.sqrt(var + 1e-3) + b

def backend(ma, mb, ypa, ypb):
    fa = np.einsum("bq,bqc->bc", softmax(ma @ w["iglooA_w_qk"]), ypa)
    fb = np.einsum("bq,bqc->bc", softmax(mb @ w["iglooB_w_qk"]), ypb)
    f = np.concatenate([fa, fb], -1)
    h1 = np.maximum(bn(f @ w["enc_dense_kernel"] + w["enc_dense_bias"], w["enc_bn_gamma"], w["enc_bn_beta"], w["enc_bn_mean"], w["enc_bn_var"]), 0)
    h2 = np.maximum(bn(h1 @ w["head_dense_kernel"] + w["head_dense_bias"], w["head_bn_gamma"], w["head_bn_beta"], w["head_bn_mean"], w["head_bn_var"]), 0)
    return softmax(h2 @ w["out_dense_kernel"] + w["out_dense_bias"])

for prec in ("f16c6", "f16c8"):
    sc, ref = [], []
    for a in range(0, n, 512):
        b = synthetic.synth_windows(a, min(512, n - a))
        sc.append(eng.classify(b, prec)); ref.append(eng.classify(b, "f32"))
    err = np.abs(np.concatenate(sc) - np.concatenate(ref)).max(1)
    worst = np.argsort(err)[-k:]
    bases = np.concatenate([synthetic.synth_windows(int(i), 1) for i in worst])
    taps = ("m_a", "m_b", "yp_a", "yp_b")
    _, t32 = eng.debug_forward(bases, "f32", taps=taps)
    _, tp = eng.debug_forward(bases, prec, taps=taps)
    t32 = {k2: v.astype(np.float64) for k2, v in t32.items()}
    tp = {k2: v.astype(np.float64) for k2, v in tp.items()}
    base = backend(t32["m_a"], t32["m_b"], t32["yp_a"], t32["yp_b"])
    rows = {
        "all four from " + prec: backend(tp["m_a"], tp["m_b"], tp["yp_a"], tp["yp_b"]),
        "only m_a": backend(tp["m_a"], t32["m_b"], t32["yp_a"], t32["yp_b"]),
        "only m_b": backend(t32["m_a"], tp["m_b"], t32["yp_a"], t32["yp_b"]),
        "only yp_a": backend(t32["m_a"], t32["m_b"], tp["yp_a"], t32["yp_b"]),
        "only yp_b": backend(t32["m_a"], t32["m_b"], t32["yp_a"], tp["yp_b"]),
    }
    print(f"{prec}: {k} worst of {n} windows (device max |dscore| vs f32 path {err[worst].max():.2e}, smallest of them {err[worst].min():.2e})")
    for name, s in rows.items():
        d = np.abs(s - base).max(1)
        print(f"   {name:22s} max {d.max():.2e}  mean {d.mean():.2e}", flush=True)
