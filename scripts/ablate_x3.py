"""What the f16x3 launch is made of: ablation builds of gnn_fused_x3.hip (scripts/mkvariant.sh x3<name> gnn_fused_x3 -DGNN_ABL_...:
parts compiled out, WRONG results by construction) timed on one box.  Usage: GENOMAD_AMD_LIB=build_variants/lib_x3nox.so ablate_x3.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from genomad_amd import _lib, synthetic  # noqa: E402
from genomad_amd.engine import NNEngine  # noqa: E402

eng = NNEngine(0, synthetic.synth_weights(), chunk=4096)
n = 16384
b, s = eng.alloc(n * 6000), eng.alloc(n * 12)
eng.synth_windows_dev(0, n, b.ptr)
for _ in range(2):
    eng.classify_dev(b.ptr, n, s.ptr, "f16x3")
eng.sync()
eng.profile_enable(True)
eng.profile_reset()
for _ in range(4):
    eng.classify_dev(b.ptr, n, s.ptr, "f16x3")
eng.sync()
ms, launches = eng.profile_get(_lib.K_FUSED)
print(f"{os.path.basename(str(_lib.LIB_PATH)):28s} fused front end {ms / launches:.3f} ms per 4096 windows", flush=True)
