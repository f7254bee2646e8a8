"""More windows for the tails gate of f16x3tk (weight seed 42): windows [first, first + n) of the synthetic set in blocks of 2^20,
max / rms of |dscore| against the exact-f32 device path per block and overall.  Usage: tails_r06_more.py [first=1048576] [n=4194304]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from genomad_amd import synthetic  # noqa: E402
from genomad_amd.engine import NNEngine  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4 << 20
blk = 1 << 20
with NNEngine(0, synthetic.synth_weights(42)) as eng:
    assert eng.build_kmer_tables()
    bases, scores = eng.alloc(blk * 6000), eng.alloc(blk * 12)
    worst, sq, cnt, over = 0.0, 0.0, 0, 0
    for a in range(first, first + n, blk):
        m = min(blk, first + n - a)
        eng.synth_windows_dev(a, m, bases.ptr)
        out = {}
        for prec in ("f32", "f16x3tk"):
            eng.classify_dev(bases.ptr, m, scores.ptr, prec)
            eng.sync()
            out[prec] = scores.download((m, 3), np.float32)
        d = np.abs(out["f16x3tk"] - out["f32"]).max(axis=1).astype(np.float64)
        worst, sq, cnt, over = max(worst, d.max()), sq + (d ** 2).sum(), cnt + m, over + int((d > 5e-5).sum())
        print(f"windows {a} .. {a + m}: max |dscore| {d.max():.3e}  rms {np.sqrt((d ** 2).mean()):.3e}  above 5e-5: {int((d > 5e-5).sum())}  finite: {bool(np.isfinite(out['f16x3tk']).all())}", flush=True)
    print(f"f16x3tk vs exact f32 over {cnt} windows [{first}, {first + n}): max {worst:.3e}  rms {np.sqrt(sq / cnt):.3e}  above 5e-5: {over}")
