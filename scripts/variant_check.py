"""A/B helper: time the fused front end, check it against the exact f32 path and print the per-phase
cycle counters, for the library selected by GENOMAD_AMD_LIB (see scripts/mkvariant.sh).
Usage: variant_check.py [n_windows]"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from genomad_amd import synthetic, _lib
from genomad_amd.engine import NNEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
eng = NNEngine(0, synthetic.synth_weights(), chunk=4096)
bases = eng.alloc(n * 6000)
scores = eng.alloc(n * 12)
eng.synth_windows_dev(0, n, bases.ptr)
eng.sync()
m = min(n, 256)
eng.classify_dev(bases.ptr, m, scores.ptr, 'f32'); eng.sync()
ref = scores.download((m, 3), np.float32)
eng.classify_dev(bases.ptr, m, scores.ptr, 'bf16x3'); eng.sync()
got = scores.download((m, 3), np.float32)
print(f"{os.environ.get('GENOMAD_AMD_LIB', 'default lib')}: max |dscore| vs f32 path on {m} windows = {np.abs(ref - got).max():.3e}")
for prec in ('bf16x3', 'bf16'):
    eng.classify_dev(bases.ptr, n, scores.ptr, prec); eng.sync()
    eng.profile_enable(True); eng.profile_reset()
    t = time.time()
    for _ in range(3):
        eng.classify_dev(bases.ptr, n, scores.ptr, prec)
    eng.sync()
    dt = (time.time() - t) / 3
    fms, fl = eng.profile_get(_lib.K_FUSED)
    bms, bl = eng.profile_get(_lib.K_BACKEND)
    print(f"  {prec}: {n / dt:.0f} windows/s; fused {fms / fl:.2f} ms per {n // (fl // 3)} windows, backend {bms / bl:.2f} ms")
    eng.profile_enable(False)

import ctypes as C
names = ["MFMA 0", "MFMA 1", "MFMA 2", "MFMA 3", "MFMA 4", "MFMA 5", "MFMA 6", "MFMA 7", "helper 8", "helper 9",
         "helper 10", "helper 11"]
_lib.check(eng.lib.gnn_phase_cycles(eng.ctx, 1, None))
eng.classify_dev(bases.ptr, n, scores.ptr, 'bf16x3'); eng.sync()
out = (C.c_uint64 * 16)()
_lib.check(eng.lib.gnn_phase_cycles(eng.ctx, 0, out))
print("  phase cycles per window-step (see GNN_TICK in gnn_fused.hip):", {nm: round(v / n / 47) for nm, v in zip(names, out)})
print("  MFMA-wave total", round(sum(out[:8]) / n / 47))
