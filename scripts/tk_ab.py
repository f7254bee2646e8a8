"""A/B timing of library variants of the k-mer-table kernel: one fresh process per library (GENOMAD_AMD_LIB), fused front end ms per
4096 windows (16 384-window launches) and the per-phase cycle counters.   python scripts/tk_ab.py lib1.so lib2.so ... [--rounds 2]"""
import os
import subprocess
import sys

WORKER = r'''
import sys, ctypes as C
import numpy as np
sys.path.insert(0, ".")
from genomad_amd import synthetic, _lib
from genomad_amd.engine import NNEngine
eng = NNEngine(0, synthetic.synth_weights(), chunk=16384)
prec = sys.argv[1]
if prec == "f16x3tk":
    assert eng.build_kmer_tables()
N = 16384
b, s = eng.alloc(N * 6000), eng.alloc(N * 12)
eng.synth_windows_dev(0, N, b.ptr)
eng.classify_dev(b.ptr, N, s.ptr, prec); eng.sync()
eng.profile_enable(True); eng.profile_reset()
for _ in range(6):
    eng.classify_dev(b.ptr, N, s.ptr, prec)
eng.sync()
ms, l = eng.profile_get(_lib.K_FUSED)
eng.profile_enable(False)
_lib.check(eng.lib.gnn_phase_cycles(eng.ctx, 1, None))
eng.classify_dev(b.ptr, 4096, s.ptr, prec); eng.sync()
out = (C.c_uint64 * 16)()
_lib.check(eng.lib.gnn_phase_cycles(eng.ctx, 0, out))
per = [v / (4096 * 63) for v in out]
print(f"{ms / l / 4:.3f} ms per 4096 windows = {N / (ms / l) * 1e3:.0f} windows/s | matrix {sum(per[:8]):.0f} " + " ".join(f"{v:.0f}" for v in per[:8]) + f" | helper {sum(per[8:]):.0f} " + " ".join(f"{v:.0f}" for v in per[8:]))
'''
libs = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--rounds"]
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 1
for r in range(rounds):
    for lib in libs:
        prec = "f16x3tk"
        if lib.endswith(":tc"):
            lib, prec = lib[:-3], "f16x3tc"
        env = dict(os.environ, GENOMAD_AMD_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", WORKER, prec], env=env, capture_output=True, text=True, timeout=600)
        print(f"{os.path.basename(lib):40s} {prec:8s} {out.stdout.strip() or out.stderr.strip()[-300:]}", flush=True)
