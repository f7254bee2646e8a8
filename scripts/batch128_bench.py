"""The reference's call pattern through the host-buffer entry point: predict() once per batch of 128 windows
(nn_classification.py:316-317) = gnn_classify(ctx, host windows, 128, ..., host scores) in a loop.  Reports calls/s and
windows/s per batch size.  Run once per library to compare (GENOMAD_AMD_LIB=build_variants/lib_r02.so = round 2, which
allocated and freed two device buffers per call).
Usage: batch128_bench.py [precision] [windows, default 8192]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from genomad_amd import _lib, synthetic  # noqa: E402
from genomad_amd.engine import NNEngine  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else _lib.DEFAULT_PRECISION
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
eng = NNEngine(0, synthetic.synth_weights())
bases = eng.synth_windows(0, n)
ref = eng.classify(bases, prec)
print(f"library {_lib.LIB_PATH}, precision {prec}, {n} windows")
for batch in (32, 128, 512, 4096):
    eng.classify(bases[:batch], prec)
    t = time.perf_counter()
    out = np.concatenate([eng.classify(bases[a:a + batch], prec) for a in range(0, n, batch)])
    dt = time.perf_counter() - t
    assert np.array_equal(out, ref), "scores depend on the batch size"
    print(f"  batch {batch:5d}: {n / batch / dt:8.1f} calls/s  {n / dt:10.1f} windows/s  ({dt / (n / batch) * 1e3:.3f} ms per call)")
eng.close()
