"""GPU check of the streaming three-pass kernel (gnn_fused_x3.hip: f16x3, bf16x3): per-stage errors against the exact f32
device path, score errors over a few thousand windows, launch time and per-phase cycles.  Run once with GNN_X3_ROUND1=1
(the round-1 kernel of gnn_fused.hip serves the two modes) for the A/B.
Usage: x3_check.py [n_windows_timed, default 16384]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from genomad_amd import _lib, synthetic  # noqa: E402
from genomad_amd.engine import NNEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
eng = NNEngine(0, synthetic.synth_weights(), chunk=4096)
which = "round-1 kernel (gnn_fused.hip)" if os.environ.get("GNN_X3_ROUND1") else "streaming kernel (gnn_fused_x3.hip)"
print("library", _lib.LIB_PATH, "|", which, flush=True)
b16 = synthetic.synth_windows(0, 16)
taps = ("m_a", "m_b", "yp_a", "yp_b", "alpha_a", "alpha_b", "feat")
s32, t32 = eng.debug_forward(b16, "f32", taps=taps)
for prec in ("f16x3", "bf16x3"):
    s, t = eng.debug_forward(b16, prec, taps=taps)
    print(prec, "vs f32 device path on 16 windows:",
          {k: f"{float(np.abs(t[k] - t32[k]).max()):.2e} (of {float(np.abs(t32[k]).max()):.2e})" for k in taps},
          "scores %.3e" % np.abs(s - s32).max(), "nan:", bool(np.isnan(s).any()), flush=True)
bases, scores = eng.alloc(n * 6000), eng.alloc(n * 12)
eng.synth_windows_dev(0, n, bases.ptr)
eng.sync()
mref = min(n, 4096)
eng.classify_dev(bases.ptr, mref, scores.ptr, "f32")
eng.sync()
ref = scores.download((mref, 3), np.float32)
for prec in ("f16x3", "bf16x3"):
    eng.classify_dev(bases.ptr, mref, scores.ptr, prec)
    eng.sync()
    d = np.abs(ref - scores.download((mref, 3), np.float32)).max(axis=1)
    print(f"{prec}: max |dscore| vs f32 device path on {mref} windows = {d.max():.3e}  (99.9th pct {np.quantile(d, 0.999):.2e})", flush=True)
for prec in ("f16x3", "bf16x3", "f16c6", "f16x3", "bf16x3", "f16c6"):
    eng.classify_dev(bases.ptr, n, scores.ptr, prec)
    eng.sync()
    eng.profile_enable(True)
    eng.profile_reset()
    t = time.time()
    for _ in range(3):
        eng.classify_dev(bases.ptr, n, scores.ptr, prec)
    eng.sync()
    dt = (time.time() - t) / 3
    fms, fl = eng.profile_get(_lib.K_FUSED)
    bms, bl = eng.profile_get(_lib.K_BACKEND)
    print(f"  {prec}: {n / dt:.0f} windows/s; fused {fms / fl:.3f} ms per {n // (fl // 3)} windows, backend {bms / bl:.3f} ms", flush=True)
    eng.profile_enable(False)
names = ["wvA", "conv2 loop", "wait B1", "conv2 epi+B2", "conv3 loop", "wait B3", "conv3 epi+B4", "wvB",
         "h pairs + table loads", "h x1 second half (B2..B3)", "h prow", "h wait B1", "h x1 first half", "h wait B2", "h wait B3", "h B3..B4"]
if not os.environ.get("GNN_X3_ROUND1"):
    for prec in ("f16x3",):
        _lib.check(eng.lib.gnn_phase_cycles(eng.ctx, 1, None))
        eng.classify_dev(bases.ptr, 4096, scores.ptr, prec)
        eng.sync()
        out = (C.c_uint64 * 16)()
        _lib.check(eng.lib.gnn_phase_cycles(eng.ctx, 0, out))
        per = [v / (4096 * 47) for v in out]
        print(f"  {prec} cycles per step (instrumented build): matrix wave total {sum(per[:8]):.0f}, helper total {sum(per[8:]):.0f}")
        print("   ", ", ".join(f"{nm} {v:.0f}" for nm, v in zip(names, per)))
# short windows: the padding skip (4096 windows of 1000 bases)
short = synthetic.synth_windows(0, 4096)
short[:, 1000:] = ord("N")
sb = eng.alloc(short.nbytes)
sb.upload(short)
for prec in ("f16x3",):
    eng.classify_dev(sb.ptr, 4096, scores.ptr, prec)
    eng.sync()
    eng.profile_enable(True)
    eng.profile_reset()
    for _ in range(3):
        eng.classify_dev(sb.ptr, 4096, scores.ptr, prec)
    eng.sync()
    fms, fl = eng.profile_get(_lib.K_FUSED)
    print(f"  {prec}: 4096 windows of 1000 bases + N padding: fused {fms / fl:.3f} ms per launch", flush=True)
    eng.profile_enable(False)
