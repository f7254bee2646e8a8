"""GPU check of the f16 + fp8-correction fused kernel (GNN_PREC_F16C8) against the exact f32 device path and
the fp64 oracle, per stage, then timing beside bf16x3 and the per-phase cycle counters.
Usage: c8_check.py [n_windows_timed]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from genomad_amd import synthetic, _lib
from genomad_amd.engine import NNEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
W = synthetic.synth_weights()
eng = NNEngine(0, W, chunk=4096)
b16 = synthetic.synth_windows(0, 16)
taps = ("m_a", "m_b", "yp_a", "yp_b", "alpha_a", "alpha_b", "feat")
s32, t32 = eng.debug_forward(b16, "f32", taps=taps)
for prec in ("bf16x3", "f16c8"):
    s, t = eng.debug_forward(b16, prec, taps=taps)
    print(prec, "vs f32 device path on 16 windows:", {k: float(np.abs(t[k] - t32[k]).max()) for k in taps},
          "scores %.3e" % np.abs(s - s32).max(), "nan:", bool(np.isnan(s).any()))
try:
    from oracle import igloo_oracle
    m = 64
    bases = synthetic.synth_windows(0, m)
    want = igloo_oracle.classify_windows(bases, W, np.float64)
    for prec in ("f32", "bf16x3", "f16c8", "f16x3"):
        got = eng.classify(bases, prec)
        print(f"{prec}: max |dscore| vs fp64 oracle on {m} windows = {np.abs(got - want).max():.3e}")
except Exception as exc:  # noqa: BLE001
    print("oracle comparison skipped:", exc)

bases = eng.alloc(n * 6000)
scores = eng.alloc(n * 12)
eng.synth_windows_dev(0, n, bases.ptr)
eng.sync()
mref = min(n, 1024)
eng.classify_dev(bases.ptr, mref, scores.ptr, 'f32'); eng.sync()
ref = scores.download((mref, 3), np.float32)
for prec in ('bf16x3', 'f16c8'):
    eng.classify_dev(bases.ptr, mref, scores.ptr, prec); eng.sync()
    got = scores.download((mref, 3), np.float32)
    print(f"{prec}: max |dscore| vs f32 device path on {mref} windows = {np.abs(ref - got).max():.3e}")
names = ["wvA", "conv2 loop", "wait B1", "conv2 epi+B2", "conv3 loop", "wait B3", "conv3 epi+B4", "wvB+wait B0",
         "helper m-partials", "helper gather"]
for prec in ('bf16x3', 'f16c8', 'f16x3', 'bf16x3', 'f16c8', 'f16x3'):
    eng.classify_dev(bases.ptr, n, scores.ptr, prec); eng.sync()
    eng.profile_enable(True); eng.profile_reset()
    t = time.time()
    for _ in range(3):
        eng.classify_dev(bases.ptr, n, scores.ptr, prec)
    eng.sync()
    dt = (time.time() - t) / 3
    fms, fl = eng.profile_get(_lib.K_FUSED)
    bms, bl = eng.profile_get(_lib.K_BACKEND)
    print(f"  {prec}: {n / dt:.0f} windows/s; fused {fms / fl:.3f} ms per {n // (fl // 3)} windows, backend {bms / bl:.3f} ms")
    eng.profile_enable(False)
for prec in ('bf16x3', 'f16c8'):
    _lib.check(eng.lib.gnn_phase_cycles(eng.ctx, 1, None))
    eng.classify_dev(bases.ptr, n, scores.ptr, prec); eng.sync()
    out = (C.c_uint64 * 16)()
    _lib.check(eng.lib.gnn_phase_cycles(eng.ctx, 0, out))
    print(f"  {prec} phase ticks per window-step:", {nm: round(v / n / 47) for nm, v in zip(names, out)},
          "matrix-wave total", round(sum(out[:8]) / n / 47))
