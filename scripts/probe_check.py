"""Print the sustained MFMA rate of gnn_mfma_probe for the library selected by GENOMAD_AMD_LIB."""
import ctypes as C, os, sys
sys.path.insert(0, '.')
from genomad_amd import synthetic, _lib
from genomad_amd.engine import NNEngine
eng = NNEngine(0, synthetic.synth_weights(), chunk=256)
for ms in (200, 1000, 1000):
    out = C.c_double()
    _lib.check(eng.lib.gnn_mfma_probe(eng.ctx, ms, C.byref(out)))
    print(os.environ.get('GENOMAD_AMD_LIB', 'default'), ms, 'ms ->', round(out.value, 1), 'TFLOP/s')
