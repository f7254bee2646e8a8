"""A/B of library variants on one box (scripts/mkvariant.sh builds them): for every library given, in alternating order, a fresh
process times the default front end (ms per 4096 windows over a 16 384-window call, 3 repeats) and hashes the scores and the
intermediates of 600 windows, so that a schedule change shows both its time and that the bits did not move.
Usage: tc_ab.py [--rounds R] lib_a.so lib_b.so ...        ("default" = the in-tree library)"""
import hashlib
import os
import subprocess
import sys

if sys.argv[1] == "--child":
    import numpy as np
    sys.path.insert(0, ".")
    from genomad_amd import synthetic, _lib
    from genomad_amd.engine import NNEngine
    eng = NNEngine(0, synthetic.synth_weights(), chunk=16384)
    N = 16384
    b, s = eng.alloc(N * 6000), eng.alloc(N * 12)
    eng.synth_windows_dev(0, N, b.ptr)
    eng.classify_dev(b.ptr, N, s.ptr, "f16x3tc")
    eng.sync()
    eng.profile_enable(True)
    eng.profile_reset()
    for _ in range(3):
        eng.classify_dev(b.ptr, N, s.ptr, "f16x3tc")
    eng.sync()
    ms, l = eng.profile_get(_lib.K_FUSED)
    eng.profile_enable(False)
    h = hashlib.md5(s.download((N, 3), np.float32).tobytes())
    wins = synthetic.synth_windows(5, 600)
    wins[7, 3000:] = 4
    wins[11, :] = 4
    sc, taps = eng.debug_forward(wins, "f16x3tc", taps=("m_a", "m_b", "yp_a", "yp_b"))
    h.update(sc.tobytes())
    for k in ("m_a", "m_b", "yp_a", "yp_b"):
        h.update(taps[k].tobytes())
    print(f"{ms / l / (N // 4096):8.3f} ms per 4096 windows   bits {h.hexdigest()[:12]}")
    sys.exit(0)

args = sys.argv[1:]
rounds = 2
if args[0] == "--rounds":
    rounds = int(args[1])
    args = args[2:]
for r in range(rounds):
    for lib in args:
        env = {k: v for k, v in os.environ.items() if k != "GENOMAD_AMD_LIB"}
        if lib != "default":
            env["GENOMAD_AMD_LIB"] = lib
        out = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True, timeout=300)
        print(f"{lib:40s} {out.stdout.strip() or out.stderr[-300:]}", flush=True)
