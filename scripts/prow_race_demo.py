"""Root-cause demonstration for the rare wrong-result mismatch recorded in profiles/history/r02c6_async_flake.md.

Round 2's f16c6 kernel wrote the conv1 pair rows of the next step (LDS, one entry per helper thread) at the top of a step
and let the OTHER helper waves' gathers read them after their pair-product loop, with no workgroup barrier in between: the
reader only came later because the loop takes a few thousand cycles.  A helper wave that reaches the step late (first touch
of a page, cold instruction cache: the mismatches were only ever seen in the first launches of fresh processes, and on
freshly allocated window buffers) lets the readers see the PREVIOUS step's pair rows for its 64 positions: 64 wrong conv1
rows in one window, scores off by 1e-3 .. 1e-1 — the recorded symptom.  This script injects that lateness on purpose
(build_variants/lib_race_*.so: helper wave 5 sleeps ~32 k cycles at the top of every 8th step, -DGNN_RACE_DELAY=4):

  round-2 kernel + delay -> wrong scores        this round's kernel (pair rows written between B3 and B4 of the
                                                 previous step: a barrier always separates writer and readers) + delay -> same bits

Usage (GPU box): python scripts/prow_race_demo.py [windows, default 2048]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048

WORKER = f"""
import sys, numpy as np
sys.path.insert(0, {ROOT!r})
from genomad_amd import synthetic
from genomad_amd.engine import NNEngine
eng = NNEngine(0, synthetic.synth_weights(), chunk=4096)
b = eng.alloc({n} * 6000); s = eng.alloc({n} * 12)
eng.synth_windows_dev(0, {n}, b.ptr)
eng.classify_dev(b.ptr, {n}, s.ptr, "f16c6"); eng.sync()
np.save(sys.argv[1], s.download(({n}, 3), np.float32))
"""

libs = [("this round's kernel", None), ("this round's kernel + delayed helper wave", "build_variants/lib_race_new.so"),
        ("round-2 kernel", "build_variants/lib_prowold.so"), ("round-2 kernel + delayed helper wave", "build_variants/lib_race_old.so")]
out = {}
for i, (name, lib) in enumerate(libs):
    env = dict(os.environ)
    if lib:
        env["GENOMAD_AMD_LIB"] = os.path.join(ROOT, lib)
    path = f"/tmp/prow_race_{i}.npy"
    subprocess.run([sys.executable, "-c", WORKER, path], env=env, check=True)
    out[name] = np.load(path)
ref = out["this round's kernel"]
for name, _ in libs[1:]:
    d = np.abs(out[name] - ref).max(axis=1)
    bad = int((out[name] != ref).any(axis=1).sum())
    print(f"{name:45s}: {bad:5d} of {n} windows differ from this round's kernel, max |dscore| {d.max():.3e}, median of the differing "
          f"{(np.median(d[d > 0]) if bad else 0.0):.3e}")
