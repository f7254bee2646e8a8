"""Rows of the 14-mer table against the exact-f32 path's x2 tap (debug aid of round 6)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from genomad_amd import synthetic, _lib  # noqa: E402
from genomad_amd.engine import NNEngine  # noqa: E402

eng = NNEngine(0, synthetic.synth_weights(), chunk=4096)
assert eng.build_kmer_tables()
wins = synthetic.synth_windows(0, 2)
ref, rt = eng.debug_forward(wins, "f32", taps=("x1", "x2", "x3"))
dig = {65: 0, 67: 1, 71: 2, 84: 3}
out = np.empty(128, np.float32)
for t in (10, 11, 50, 1000, 5996):
    code = 0
    for b in wins[0, t - 10:t + 4]:
        code = code * 4 + dig[int(b)]
    _lib.check(eng.lib.gnn_debug_kmer_table_row(eng.ctx, 0, code, out.ctypes.data))
    d = np.abs(out - rt["x2"][0, t])
    print(f"t={t}: 14-mer {code}: max|table - x2 tap| {d.max():.3e}  (|x2| max {np.abs(rt['x2'][0, t]).max():.3f}) first values {out[:4]} vs {rt['x2'][0, t][:4]}")
