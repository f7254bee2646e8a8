"""Stress of gnn_classify_dev_async beside synchronous entry points: many in-process repetitions of the mix the GPU test
test_asynchronous_classification_is_bit_identical runs once, in variants that isolate what a mismatch depends on (tiny calls,
a debug forward in between).  Prints mismatch counts and where the rows differ.
Usage: async_stress.py [iterations per variant] [variant ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from genomad_amd import synthetic
from genomad_amd.engine import NNEngine

args = sys.argv[1:]
iters = int(args[0]) if args else 40
only = args[1:]
W = synthetic.synth_weights()
eng = NNEngine(0, W)
n = 6 * 1024 + 300
bases, a, b, c = eng.alloc(n * 6000), eng.alloc(n * 12), eng.alloc(n * 12), eng.alloc(64 * 12)
eng.synth_windows_dev(4242, n, bases.ptr)
eng.classify_dev(bases.ptr, n, a.ptr, "f16c6")
eng.sync()
want = a.download((n, 3), np.float32)
eng.classify_dev(bases.ptr, n, a.ptr, "f16c6")
eng.sync()
assert np.array_equal(want, a.download((n, 3), np.float32)), "synchronous path is not deterministic"
tb = synthetic.synth_windows(7, 4)
s_ref, t_ref = eng.debug_forward(tb, "f16c6")
tb_dev = eng.alloc(4 * 6000)
tb_dev.upload(tb)
nan = np.full((n, 3), np.nan, np.float32)
TINY = [0, 1024, 1030, 2048, 3072, 3073, 5000, 6144, n]
ROUND = [0, 1024, 2048, 3072, 4096, 5120, 6144, n]


def ranges(idx):
    out, start, prev = [], None, None
    for i in idx:
        if start is None:
            start = prev = i
        elif i == prev + 1:
            prev = i
        else:
            out.append((start, prev))
            start = prev = i
    if start is not None:
        out.append((start, prev))
    return out


def run(cuts, middle):
    bad_main, bad_mid, notes = 0, 0, []
    for it in range(iters):
        b.upload(nan)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            eng.classify_dev_async(bases.ptr + lo * 6000, hi - lo, b.ptr + lo * 12, "f16c6")
            if lo == 2048:
                if middle == "debug":
                    s1, t1 = eng.debug_forward(tb, "f16c6")
                    s2, t2 = eng.debug_forward(tb, "f16c6")
                    w1 = [k for k in t1 if not np.array_equal(t1[k], t_ref[k])]
                    w2 = [k for k in t2 if not np.array_equal(t2[k], t_ref[k])]
                    if not (np.array_equal(s1, s_ref) and np.array_equal(s2, s_ref)) or w1 or w2:
                        bad_mid += 1
                        notes.append(f"it {it}: debug 1 rows {np.nonzero((s1 != s_ref).any(axis=1))[0].tolist()} taps {w1}; "
                                     f"debug 2 rows {np.nonzero((s2 != s_ref).any(axis=1))[0].tolist()} taps {w2}")
                elif middle == "sync4":
                    for _ in range(2):
                        eng.classify_dev(tb_dev.ptr, 4, c.ptr, "f16c6")
                        eng.sync()
                        s = c.download((4, 3), np.float32)
                        if not np.array_equal(s, s_ref):
                            bad_mid += 1
                            notes.append(f"it {it}: sync4 rows {np.nonzero((s != s_ref).any(axis=1))[0].tolist()}")
        eng.flush()
        eng.sync()
        got = b.download((n, 3), np.float32)
        d = np.nonzero(~(got == want).all(axis=1))[0]
        if len(d):
            bad_main += 1
            notes.append(f"it {it}: async rows {ranges(d.tolist())[:8]} ({len(d)} rows, {int(np.isnan(got[d]).any(axis=1).sum())} with NaN)")
    return bad_main, bad_mid, notes


VARIANTS = {
    "full": (TINY, "debug"),
    "no_middle": (TINY, None),
    "sync4_middle": (TINY, "sync4"),
    "round_cuts_debug": (ROUND, "debug"),
    "round_cuts_no_middle": (ROUND, None),
}
for name, (cuts, middle) in VARIANTS.items():
    if only and name not in only:
        continue
    t = time.time()
    bm, bd, notes = run(cuts, middle)
    print(f"{name}: {bm}/{iters} async mismatches, {bd} middle mismatches  ({time.time() - t:.1f} s)", flush=True)
    for s in notes[:6]:
        print("    " + s, flush=True)
