"""Time the f16c6 fused kernel of the library selected by GENOMAD_AMD_LIB (ablation builds give wrong results;
only the time matters).  Usage: ablate_c6.py [n_windows]"""
import os, sys
sys.path.insert(0, '.')
from genomad_amd import synthetic, _lib
from genomad_amd.engine import NNEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
eng = NNEngine(0, synthetic.synth_weights(), chunk=4096)
bases, scores = eng.alloc(n * 6000), eng.alloc(n * 12)
eng.synth_windows_dev(0, n, bases.ptr); eng.sync()
eng.classify_dev(bases.ptr, n, scores.ptr, 'f16c6'); eng.sync()
eng.profile_enable(True); eng.profile_reset()
for _ in range(4):
    eng.classify_dev(bases.ptr, n, scores.ptr, 'f16c6')
eng.sync()
fms, fl = eng.profile_get(_lib.K_FUSED)
print(f"{os.environ.get('GENOMAD_AMD_LIB', 'default')} rows/step {eng.lib.gnn_fused_rows_per_step(5)}: fused {fms / fl:.3f} ms per 4096 windows", flush=True)
