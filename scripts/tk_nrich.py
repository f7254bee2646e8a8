"""How f16x3tk and f16x3tc behave on windows full of rows no 14-mer indexes (scattered non-ACGT bytes): ms per 1024 windows."""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from genomad_amd import synthetic  # noqa: E402
from genomad_amd.engine import NNEngine  # noqa: E402

eng = NNEngine(0, synthetic.synth_weights(), chunk=4096)
assert eng.build_kmer_tables()
rng = np.random.default_rng(3)
n = 1024
for frac in (0.0, 0.001, 0.01, 0.05):
    w = synthetic.synth_windows(0, n).copy()
    mask = rng.random(w.shape) < frac
    w[mask] = ord("N")
    b, s = eng.alloc(n * 6000), eng.alloc(n * 12)
    b.upload(w)
    out = {}
    for prec in ("f16x3tc", "f16x3tk"):
        eng.classify_dev(b.ptr, n, s.ptr, prec); eng.sync()
        t = time.perf_counter()
        for _ in range(3):
            eng.classify_dev(b.ptr, n, s.ptr, prec)
        eng.sync()
        out[prec] = ((time.perf_counter() - t) / 3 * 1e3, s.download((n, 3), np.float32))
    eng.classify_dev(b.ptr, n, s.ptr, "f32"); eng.sync()
    ex = s.download((n, 3), np.float32)
    print(f"N fraction {frac}: f16x3tc {out['f16x3tc'][0]:.2f} ms, f16x3tk {out['f16x3tk'][0]:.2f} ms per {n} windows; max|tk - f32| {np.abs(out['f16x3tk'][1] - ex).max():.2e}, max|tc - f32| {np.abs(out['f16x3tc'][1] - ex).max():.2e}")
    b.free(); s.free()
