"""Max |dscore| of the fused paths against the exact f32 device path on n synthetic windows, for several weight
seeds (the f32 device path itself sits within ~4e-6 of the fp64 oracle).  Usage: seed_check.py [n] [seeds...]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from genomad_amd import synthetic
from genomad_amd.engine import NNEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
seeds = [int(s) for s in sys.argv[2:]] or [42, 43, 7]
for seed in seeds:
    with NNEngine(0, synthetic.synth_weights(seed)) as eng:
        bases, scores = eng.alloc(n * 6000), eng.alloc(n * 12)
        eng.synth_windows_dev(0, n, bases.ptr)
        out = {}
        for prec in ("f32", "bf16x3", "f16c8", "f16c6", "f16x3"):
            eng.classify_dev(bases.ptr, n, scores.ptr, prec)
            eng.sync()
            out[prec] = scores.download((n, 3), np.float32)
        print(f"seed {seed}: n={n}  score std per class {out['f32'].std(0).round(3).tolist()}  "
              f"bf16x3 vs f32 {np.abs(out['bf16x3'] - out['f32']).max():.3e}  "
              f"f16c8 vs f32 {np.abs(out['f16c8'] - out['f32']).max():.3e}  "
              f"f16c6 vs f32 {np.abs(out['f16c6'] - out['f32']).max():.3e}  "
              f"f16x3 vs f32 {np.abs(out['f16x3'] - out['f32']).max():.3e}  "
              f"(99.9th pct f16c8 {np.quantile(np.abs(out['f16c8'] - out['f32']).max(1), 0.999):.3e}, "
              f"f16c6 {np.quantile(np.abs(out['f16c6'] - out['f32']).max(1), 0.999):.3e})", flush=True)
