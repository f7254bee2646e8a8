"""Tails of the class-score error of the k-mer-table arithmetic (f16x3tk) beside the default (f16x3tc) against the exact-f32 device path:
10 000 / 100 000 / 1 048 576 synthetic windows (prefixes of one run), weight seeds 42 and 43 (43 with the output bias re-centred so
that all three classes vary: tests/test_gpu_parity.py::_calibrated_weights) - round 4's gate (scripts/tails_r04.py) for round 6's kernel.
Usage: tails_r06.py [n, default 1048576]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from genomad_amd import synthetic  # noqa: E402
from genomad_amd.engine import NNEngine  # noqa: E402
from tests.test_gpu_parity import _calibrated_weights  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
for seed in (42, 43):
    w = synthetic.synth_weights(42) if seed == 42 else _calibrated_weights(43)
    with NNEngine(0, w) as eng:
        assert eng.build_kmer_tables()
        bases, scores = eng.alloc(n * 6000), eng.alloc(n * 12)
        eng.synth_windows_dev(0, n, bases.ptr)
        out = {}
        for prec in ("f32", "f16x3tc", "f16x3tk"):
            eng.classify_dev(bases.ptr, n, scores.ptr, prec)
            eng.sync()
            out[prec] = scores.download((n, 3), np.float32)
        print(f"weight seed {seed}: score std per class {out['f32'].std(0).round(3).tolist()}", flush=True)
        for prec in ("f16x3tc", "f16x3tk"):
            d = np.abs(out[prec] - out["f32"]).max(axis=1)
            for m in (10_000, 100_000, n):
                if m > n:
                    continue
                dm = d[:m]
                print(f"  {prec:8s} {m:8d} windows: max |dscore| {dm.max():.3e}  99.9th pct {np.quantile(dm, 0.999):.3e}  "
                      f"rms {np.sqrt((dm.astype(np.float64) ** 2).mean()):.3e}  above 5e-5: {int((dm > 5e-5).sum())}  above 1e-4: {int((dm > 1e-4).sum())}", flush=True)
        d = np.abs(out["f16x3tk"] - out["f16x3tc"]).max(axis=1)
        print(f"  f16x3tk vs f16x3tc: max {d.max():.3e}  99.9th pct {np.quantile(d, 0.999):.3e}", flush=True)
