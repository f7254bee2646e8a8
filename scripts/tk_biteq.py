"""Bit identity of two builds of the k-mer-table kernel (a change of the schedule or of the way rows are requested must not change one bit):
scores of 2 x 262 144 seeded windows and scores + intermediates of 2 048 mixed ones (scattered N at 0 / 0.1 / 1 / 5 %, N runs, N tails, N heads,
lower case) -> npz, or compared with one.

    GENOMAD_AMD_LIB=build_variants/lib_old.so python scripts/tk_biteq.py ref /tmp/biteq.npz     (scripts/mkvariant.sh builds variants)
    python scripts/tk_biteq.py cmp /tmp/biteq.npz
"""
import sys
import numpy as np
sys.path.insert(0, ".")
from genomad_amd import synthetic, _lib
from genomad_amd.engine import NNEngine
mode, path = sys.argv[1], sys.argv[2]
eng = NNEngine(0, synthetic.synth_weights())
assert eng.build_kmer_tables()
out = {}
N = 262144
b, s = eng.alloc(N * 6000), eng.alloc(N * 12)
for seed in (0, 5_000_000):
    eng.synth_windows_dev(seed, N, b.ptr)
    eng.classify_dev(b.ptr, N, s.ptr, "f16x3tk"); eng.sync()
    out[f"bulk{seed}"] = s.download((N, 3), np.float32)
rng = np.random.default_rng(7)
w = synthetic.synth_windows(9000, 2048).copy()
for i in range(2048):
    f = (0.0, 0.001, 0.01, 0.05)[i & 3]
    if f:
        w[i][rng.random(6000) < f] = ord("N")
    if i % 16 == 5:
        a = int(rng.integers(0, 5900)); w[i][a:a + int(rng.integers(1, 400))] = ord("N")
    if i % 16 == 9:
        w[i][int(rng.integers(100, 6000)):] = ord("N")
    if i % 16 == 11:
        w[i][:int(rng.integers(1, 30))] = ord("N")
    if i % 32 == 7:
        w[i][100:200] += 32
taps = ("m_a", "m_b", "yp_a", "yp_b")
for k0 in range(0, 2048, 512):
    sc, tp = eng.debug_forward(w[k0:k0 + 512], "f16x3tk", taps=taps)
    out[f"s{k0}"] = sc
    for k in taps:
        out[f"{k}{k0}"] = tp[k]
print(mode, _lib.LIB_PATH, len(out), "arrays")
if mode == "ref":
    np.savez(path, **out)
else:
    ref = np.load(path)
    bad = [k for k in out if not np.array_equal(out[k].view(np.uint32), ref[k].view(np.uint32))]
    print("MISMATCH: " + ", ".join(bad) if bad else f"OK: {len(out)} arrays ({2 * N} bulk windows' scores, 2048 mixed windows' scores and taps) bit-identical")
