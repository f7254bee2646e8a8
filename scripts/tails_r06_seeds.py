"""scripts/tails_r06.py for further weight sets: tails of the class-score error of f16x3tk (and f16x3tc) against the exact-f32 device path over n
synthetic windows per weight seed (output bias re-centred per seed: tests/test_gpu_parity.py::_calibrated_weights).
Usage: tails_r06_seeds.py n seed [seed ...]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from genomad_amd.engine import NNEngine  # noqa: E402
from tests.test_gpu_parity import _calibrated_weights  # noqa: E402

n = int(sys.argv[1])
for seed in map(int, sys.argv[2:]):
    w = _calibrated_weights(seed)
    with NNEngine(0, w) as eng:
        assert eng.build_kmer_tables()
        bases, scores = eng.alloc(n * 6000), eng.alloc(n * 12)
        eng.synth_windows_dev(seed * 10_000_000, n, bases.ptr)       # other windows than seeds 42 / 43 saw
        out = {}
        for prec in ("f32", "f16x3tc", "f16x3tk"):
            eng.classify_dev(bases.ptr, n, scores.ptr, prec)
            eng.sync()
            out[prec] = scores.download((n, 3), np.float32)
        print(f"weight seed {seed}: score std per class {out['f32'].std(0).round(3).tolist()}", flush=True)
        for prec in ("f16x3tc", "f16x3tk"):
            d = np.abs(out[prec] - out["f32"]).max(axis=1)
            print(f"  {prec:8s} {n:9d} windows: max |dscore| {d.max():.3e}  99.9th pct {np.percentile(d, 99.9):.3e}  rms {np.sqrt((d.astype(np.float64) ** 2).mean()):.3e}"
                  f"  above 5e-5: {int((d > 5e-5).sum())}  above 1e-4: {int((d > 1e-4).sum())}  finite: {bool(np.isfinite(out[prec]).all())}", flush=True)
