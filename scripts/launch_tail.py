"""Pricing of a persistent grid without building one: what would a workgroup that loops over windows save?

The default kernel runs one workgroup per window, one workgroup per CU (157 KB of LDS), so a launch of n windows is n / 256 "rounds"
that the hardware dispatcher fills dynamically.  A persistent grid can remove (a) the per-workgroup start (dispatch, LDS allocation,
argument loads, the first step's prologue) and (b) the quantisation of the last round.  Both are bounded from the launch time as a
function of n: T(n) = a + b n / 256 for equal-cost windows (padding skip off), and T(n) / n with the benchmark's mix of window
lengths (padding skip on: every 16th window is shorter).  HIP events of the library around the fused kernel, 5 launches per size.

    python scripts/launch_tail.py            (one GPU, ~40 s)"""
import sys

import numpy as np

sys.path.insert(0, ".")
from genomad_amd import _lib, synthetic  # noqa: E402
from genomad_amd.engine import NNEngine  # noqa: E402

eng = NNEngine(0, synthetic.synth_weights(), chunk=32768)
NMAX = 16384 + 2048
b, s = eng.alloc(NMAX * 6000), eng.alloc(NMAX * 12)
eng.synth_windows_dev(0, NMAX, b.ptr)
eng.classify_dev(b.ptr, 16384, s.ptr, "f16x3tc")
eng.sync()
sizes = (256, 512, 1024, 2048, 4096, 8192, 16384, 16384 + 64, 16384 + 256, 16384 + 2048)
for skip in (0, 1):
    _lib.check(eng.lib.gnn_debug_set_pad_skip(eng.ctx, skip))
    print(f"padding skip {'on (benchmark mix: every 16th window shorter)' if skip else 'off (equal-cost windows)'}")
    rows = []
    for n in sizes:
        eng.classify_dev(b.ptr, n, s.ptr, "f16x3tc")
        eng.sync()
        eng.profile_enable(True)
        eng.profile_reset()
        for _ in range(5):
            eng.classify_dev(b.ptr, n, s.ptr, "f16x3tc")
        ms, l = eng.profile_get(_lib.K_FUSED)
        eng.profile_enable(False)
        t = ms / l
        rows.append((n, t))
        print(f"  n = {n:6d} ({n / 256:6.2f} rounds)   {t:9.4f} ms   {t / (n / 256):8.5f} ms per round   {t / n * 4096:8.4f} ms per 4096 windows")
    x = np.array([r[0] / 256 for r in rows[:7]])
    y = np.array([r[1] for r in rows[:7]])
    bfit, afit = np.polyfit(x, y, 1)
    print(f"  fit over 1 .. 64 rounds: T = {afit:.4f} ms + {bfit:.5f} ms x rounds; the fixed part is {afit / rows[6][1] * 100:.2f} % of a "
          f"16 384-window launch; one round alone takes {rows[0][1] / bfit:.3f} x the marginal round")
    print(f"  a part-filled last round: +64 windows cost {rows[7][1] - rows[6][1]:.4f} ms, +256 cost {rows[8][1] - rows[6][1]:.4f} ms "
          f"(marginal round {bfit:.4f} ms)")
