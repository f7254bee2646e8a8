"""Second stress of synchronous host-buffer forwards beside pending asynchronous classifications: host-side delays of
0-10 ms between the asynchronous calls and the forward, two alternating window sets (so that a stale buffer shows up as the
other set's scores), with and without a host copy of the bases.
Usage: async_stress2.py [iterations]"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from genomad_amd import synthetic
from genomad_amd.engine import NNEngine

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
W = synthetic.synth_weights()
eng = NNEngine(0, W)
n = 4096
bases, b, c = eng.alloc(n * 6000), eng.alloc(n * 12), eng.alloc(64 * 12)
eng.synth_windows_dev(4242, n, bases.ptr)
eng.classify_dev(bases.ptr, n, b.ptr, "f16c6")
eng.sync()
want = b.download((n, 3), np.float32)
tbs = [synthetic.synth_windows(7, 4), synthetic.synth_windows(300, 4)]
refs = [eng.debug_forward(t, "f16c6") for t in tbs]
tb_dev = [eng.alloc(4 * 6000), eng.alloc(4 * 6000)]
for d, t in zip(tb_dev, tbs):
    d.upload(t)
DELAYS = [0.0, 0.0003, 0.001, 0.003, 0.01]


def describe(s, t, k):
    rows = np.nonzero((s != refs[k][0]).any(axis=1))[0].tolist()
    other = [r for r in rows if np.array_equal(s[r], refs[1 - k][0][r])]
    taps = [name for name in t if not np.array_equal(t[name], refs[k][1][name])] if t else []
    return f"rows {rows} (equal to the OTHER set's scores: {other}) taps {taps}"


def run(name, use_async, mode):
    bad = {d: 0 for d in DELAYS}
    bad_async = 0
    notes = []
    t0 = time.time()
    for it in range(iters):
        k = it & 1
        delay = DELAYS[it % len(DELAYS)]
        if use_async:
            for lo in range(0, n, 1024):
                eng.classify_dev_async(bases.ptr + lo * 6000, 1024, b.ptr + lo * 12, "f16c6")
        if delay:
            time.sleep(delay)
        if mode == "host":
            s, t = eng.debug_forward(tbs[k], "f16c6")
        else:
            eng.classify_dev(tb_dev[k].ptr, 4, c.ptr, "f16c6")
            eng.sync()
            s, t = c.download((4, 3), np.float32), None
        if not np.array_equal(s, refs[k][0]) or (t and any(not np.array_equal(t[x], refs[k][1][x]) for x in t)):
            bad[delay] += 1
            notes.append(f"it {it} delay {delay * 1e3:.1f} ms: " + describe(s, t, k))
        if use_async:
            eng.flush()
            eng.sync()
            got = b.download((n, 3), np.float32)
            if not np.array_equal(got, want):
                bad_async += 1
                d = np.nonzero(~(got == want).all(axis=1))[0]
                notes.append(f"it {it}: async rows {d[:6].tolist()}.. ({len(d)})")
    print(f"{name}: forward mismatches per delay {dict((f'{d * 1e3:.1f}ms', v) for d, v in bad.items())}, async mismatches {bad_async}/{iters}"
          f"  ({time.time() - t0:.1f} s)", flush=True)
    for s in notes[:5]:
        print("    " + s, flush=True)


run("host forward alone", False, "host")
run("host forward beside async", True, "host")
run("device forward beside async", True, "dev")
