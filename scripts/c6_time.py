import sys, os
sys.path.insert(0, ".")
from genomad_amd import _lib, synthetic
from genomad_amd.engine import NNEngine
eng = NNEngine(0, synthetic.synth_weights(), chunk=4096)
n = 16384
b, s = eng.alloc(n * 6000), eng.alloc(n * 12)
eng.synth_windows_dev(0, n, b.ptr)
for _ in range(2): eng.classify_dev(b.ptr, n, s.ptr, "f16c6")
eng.sync(); eng.profile_enable(True); eng.profile_reset()
for _ in range(4): eng.classify_dev(b.ptr, n, s.ptr, "f16c6")
eng.sync(); ms, l = eng.profile_get(_lib.K_FUSED)
print(f"{os.path.basename(str(_lib.LIB_PATH)):28s} f16c6 fused front end {ms / l:.3f} ms per 4096 windows", flush=True)
