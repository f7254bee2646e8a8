"""One FRESH-PROCESS sample of the call mix in which gnn_classify_dev_async once mismatched (profiles/history/r02c6_async_flake.md):
the mismatch showed up in about 1 of 17 fresh processes and never in > 800 in-process repetitions, so the unit of this
hunt is a process.  Prints one line: "OK ..." or "FAIL <what differed, where>".  scripts/async_hunt.sh runs it under one
runtime / library setting at a time.

Scenario (the round-2 test as it was when it failed, repeated `rounds` times in the process):
  synchronous classification of 6444 windows -> want
  asynchronous calls over the cuts [0, 1024, 1030, 2048, 3072 | two tapped host forwards of 4 windows | 3073, 5000, 6144, n]
  flush, compare with want; a third forward after everything drained is the reference for the two in between.
Usage: async_hunt.py [rounds, default 2] [precision, default f16c6]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from genomad_amd import synthetic  # noqa: E402
from genomad_amd.engine import NNEngine  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prec = sys.argv[2] if len(sys.argv) > 2 else "f16c6"
t_start = time.time()
cache = "/tmp/gnn_hunt_weights.npz"
if os.path.exists(cache):
    W = dict(np.load(cache))
else:
    W = synthetic.synth_weights()
    tmp = f"{cache}.{os.getpid()}.tmp.npz"                 # several fresh processes may start at once (tests run three at a time)
    np.savez(tmp, **W)
    os.replace(tmp, cache)
eng = NNEngine(0, W)
n = 6 * 1024 + 300
bases, a, b = eng.alloc(n * 6000), eng.alloc(n * 12), eng.alloc(n * 12)
eng.synth_windows_dev(4242, n, bases.ptr)
eng.classify_dev(bases.ptr, n, a.ptr, prec)
eng.sync()
want = a.download((n, 3), np.float32)
cuts = [0, 1024, 1030, 2048, 3072, 3073, 5000, 6144, n]
nan = np.full((n, 3), np.nan, np.float32)
report = []


def rows(x, y):
    return np.nonzero((x != y).reshape(len(x), -1).any(axis=1))[0].tolist()


def where(k, x, y):
    """compact description of where tap k differs"""
    d = np.argwhere(x != y)
    out = {"tap": k, "values": int(len(d)), "windows": sorted(set(d[:, 0].tolist())), "max": float(np.abs(x - y).max())}
    if x.ndim == 3:      # yp: (window, pooled row, channel)
        out["rows"] = sorted(set(d[:, 1].tolist()))[:24]
        out["channels"] = sorted(set((d[:, 2] // 32).tolist()))
    elif x.ndim == 2:
        out["cols"] = sorted(set(d[:, 1].tolist()))[:24]
    out["nan"] = int(np.isnan(x).sum())
    return out


for rnd in range(rounds):
    b.upload(nan)
    mid = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        eng.classify_dev_async(bases.ptr + lo * 6000, hi - lo, b.ptr + lo * 12, prec)
        if lo == 2048:
            tb = synthetic.synth_windows(7, 4)
            mid = [eng.debug_forward(tb, prec), eng.debug_forward(tb, prec)]
    eng.flush()
    eng.sync()
    got = b.download((n, 3), np.float32)
    s3, t3 = eng.debug_forward(tb, prec)
    for i, (s, t) in enumerate(mid):
        if not np.array_equal(s, s3, equal_nan=False):
            report.append({"round": rnd, "forward": i + 1, "score_rows": rows(s, s3), "max": float(np.nanmax(np.abs(s - s3))),
                           "taps": [where(k, t[k], t3[k]) for k in t if not np.array_equal(t[k], t3[k])]})
        elif any(not np.array_equal(t[k], t3[k]) for k in t):
            report.append({"round": rnd, "forward": i + 1, "score_rows": [],
                           "taps": [where(k, t[k], t3[k]) for k in t if not np.array_equal(t[k], t3[k])]})
    bad = rows(got, want)
    if bad:
        report.append({"round": rnd, "async_rows": len(bad), "first": bad[:16], "last": bad[-4:], "mod8": sorted(set(r % 8 for r in bad)),
                       "max": float(np.nanmax(np.abs(got - want)[bad])), "nan_rows": int(np.isnan(got[bad]).any(axis=1).sum())})
for buf in (bases, a, b):
    buf.free()
eng.close()
dt = time.time() - t_start
print(("FAIL " + json.dumps(report)) if report else f"OK {dt:.2f}s", flush=True)
