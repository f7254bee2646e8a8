"""Delay-injection check of the Toom-Cook kernel's barrier schedule (the method that made round 2's pair-row race deterministic):
genomad_amd/csrc/libgenomad_nn_hip_jitter.so (built by build.sh) is the library with -DTC_JITTER - every wave of the fused kernel sleeps a pseudo-random time behind
every barrier.  LDS producer / consumer pairs that a barrier orders do not care; an unordered pair shows up as a bit mismatch.

    python scripts/tc_jitter_check.py ref  out.npz          (normal library)
    GENOMAD_AMD_LIB=genomad_amd/csrc/libgenomad_nn_hip_jitter.so python scripts/tc_jitter_check.py cmp out.npz
"""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from genomad_amd import synthetic, _lib  # noqa: E402
from genomad_amd.engine import NNEngine  # noqa: E402

mode, path = sys.argv[1], sys.argv[2]
prec = sys.argv[3] if len(sys.argv) > 3 else "f16x3tc"       # "f16x3tk": the k-mer-table kernel (9 barriers per step), same method
eng = NNEngine(0, synthetic.synth_weights())
if prec == "f16x3tk" and not eng.build_kmer_tables():
    print("the device cannot hold the k-mer tables")
    sys.exit(77)
wins = synthetic.synth_windows(7000, 600)
wins[3] = np.frombuffer(b"N" * 6000, np.uint8)
wins[4, 900:] = ord("N")
wins[5, 2000:2030] = ord("N")            # an N run inside: rows no 14-mer indexes (f16x3tk: the tap-table path)
wins[6, 300:310] += 32                  # lower case
taps = ("m_a", "m_b", "yp_a", "yp_b")
out = {}
t = time.time()
for rep in range(3):                       # different sleeps every launch (the hash is salted with the workgroup index only: same
    for n in (600, 128, 40):               # launch shape -> same sleeps; different shapes and the time split vary them)
        s, tp = eng.debug_forward(wins[:n], prec, taps=taps)
        out[f"s{n}"] = s
        for k in taps:
            out[f"{k}{n}"] = tp[k]
print(f"{mode}: library {_lib.LIB_PATH}, {time.time() - t:.2f} s for 9 forwards")
if mode == "ref":
    np.savez(path, **out)
else:
    ref = np.load(path)
    bad = [k for k in out if not np.array_equal(out[k], ref[k])]
    print("MISMATCH in " + ", ".join(bad) if bad else f"OK: {len(out)} arrays bit-identical to the normal library under delay injection")
    sys.exit(1 if bad else 0)
