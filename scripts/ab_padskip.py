"""A/B of the f16c6 padding skip inside one process (same box, same clocks): ab_padskip.py [n_windows] [windows per launch]"""
import sys
sys.path.insert(0, '.')
from genomad_amd import synthetic, _lib
from genomad_amd.engine import NNEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
eng = NNEngine(0, synthetic.synth_weights(), chunk=chunk)
bases, scores = eng.alloc(n * 6000), eng.alloc(n * 12)
eng.synth_windows_dev(0, n, bases.ptr); eng.sync()
for rnd in range(3):
    for on in (0, 1):
        _lib.check(eng.lib.gnn_debug_set_pad_skip(eng.ctx, on))
        eng.classify_dev(bases.ptr, n, scores.ptr, 'f16c6'); eng.sync()
        eng.profile_enable(True); eng.profile_reset()
        for _ in range(4):
            eng.classify_dev(bases.ptr, n, scores.ptr, 'f16c6')
        eng.sync()
        fms, fl = eng.profile_get(_lib.K_FUSED)
        eng.profile_enable(False)
        print(f"pad skip {'on ' if on else 'off'}: fused {fms / fl * 4096 / chunk:.3f} ms per 4096 windows (launches of {chunk})", flush=True)
