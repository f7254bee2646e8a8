"""GPU check of the f16 + MX-fp6-correction fused kernel (GNN_PREC_F16C6) against the exact f32 device path and the
fp64 oracle, per stage, then timing beside f16c8.
Usage: c6_check.py [n_windows_timed] [--no-oracle]"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from genomad_amd import synthetic, _lib
from genomad_amd.engine import NNEngine

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 8192
W = synthetic.synth_weights()
eng = NNEngine(0, W, chunk=4096)
print('library', _lib.LIB_PATH, 'f16c6 rows per step', eng.lib.gnn_fused_rows_per_step(_lib.PRECISIONS['f16c6']), flush=True)
b16 = synthetic.synth_windows(0, 16)
taps = ("m_a", "m_b", "yp_a", "yp_b", "alpha_a", "alpha_b", "feat")
s32, t32 = eng.debug_forward(b16, "f32", taps=taps)
for prec in ("f16c8", "f16c6"):
    s, t = eng.debug_forward(b16, prec, taps=taps)
    err = {k: float(np.abs(t[k] - t32[k]).max()) for k in taps}
    scale = {k: float(np.abs(t32[k]).max()) for k in taps}
    print(prec, "vs f32 device path on 16 windows:", {k: f"{err[k]:.2e} (of {scale[k]:.2e})" for k in taps},
          "scores %.3e" % np.abs(s - s32).max(), "nan:", bool(np.isnan(s).any()), flush=True)
    if prec == "f16c6" and not np.isfinite(s).all():
        bad = {k: int((~np.isfinite(t[k])).sum()) for k in taps}
        print("  non-finite counts per tap:", bad)
if "--no-oracle" not in sys.argv:
    from oracle import igloo_oracle
    m = 64
    bases = synthetic.synth_windows(0, m)
    want = igloo_oracle.classify_windows(bases, W, np.float64)
    for prec in ("f32", "f16c8", "f16c6"):
        got = eng.classify(bases, prec)
        print(f"{prec}: max |dscore| vs fp64 oracle on {m} windows = {np.abs(got - want).max():.3e}", flush=True)

bases = eng.alloc(n * 6000)
scores = eng.alloc(n * 12)
eng.synth_windows_dev(0, n, bases.ptr)
eng.sync()
mref = min(n, 2048)
eng.classify_dev(bases.ptr, mref, scores.ptr, 'f32'); eng.sync()
ref = scores.download((mref, 3), np.float32)
for prec in ('f16c8', 'f16c6'):
    eng.classify_dev(bases.ptr, mref, scores.ptr, prec); eng.sync()
    got = scores.download((mref, 3), np.float32)
    d = np.abs(ref - got).max(axis=1)
    print(f"{prec}: max |dscore| vs f32 device path on {mref} windows = {d.max():.3e}  (99.9th pct {np.quantile(d, 0.999):.2e})", flush=True)
for prec in ('f16c8', 'f16c6', 'f16c8', 'f16c6'):
    eng.classify_dev(bases.ptr, n, scores.ptr, prec); eng.sync()
    eng.profile_enable(True); eng.profile_reset()
    t = time.time()
    for _ in range(3):
        eng.classify_dev(bases.ptr, n, scores.ptr, prec)
    eng.sync()
    dt = (time.time() - t) / 3
    fms, fl = eng.profile_get(_lib.K_FUSED)
    bms, bl = eng.profile_get(_lib.K_BACKEND)
    print(f"  {prec}: {n / dt:.0f} windows/s; fused {fms / fl:.3f} ms per {n // (fl // 3)} windows, backend {bms / bl:.3f} ms", flush=True)
    eng.profile_enable(False)
import ctypes as C
names = ["wvA", "conv2 loop", "wait B1", "conv2 epi+B2", "conv3 loop", "wait B3", "conv3 epi+B4", "wvB",
         "h pairs + table loads", "h x1 second half (B2..B3)", "h prow", "h wait B1", "h x1 first half", "h wait B2", "h wait B3", "h B3..B4"]
names8 = names[:8] + ["helper pairs (B4..B1)", "helper B1..B4"]
for prec in ('f16c8', 'f16c6'):
    rows = eng.lib.gnn_fused_rows_per_step(_lib.PRECISIONS[prec])
    steps = (5997 + rows - 1) // rows
    _lib.check(eng.lib.gnn_phase_cycles(eng.ctx, 1, None))
    eng.classify_dev(bases.ptr, n, scores.ptr, prec); eng.sync()
    out = (C.c_uint64 * 16)()
    _lib.check(eng.lib.gnn_phase_cycles(eng.ctx, 0, out))
    print(f"  {prec} phase cycles (s_memtime) per window-step of {rows} rows:", {nm: round(v / n / steps) for nm, v in zip(names if prec == 'f16c6' else names8, out)},
          "matrix-wave total", round(sum(out[:8]) / n / steps), "per 128 rows", round(sum(out[:8]) / n / steps * 128 / rows), flush=True)
