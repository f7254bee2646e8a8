"""First-light / regression check of the Toom-Cook front end (GNN_PREC_F16X3TC) on the GPU box: intermediates and scores against the
default arithmetic and the exact-f32 path, timing of the fused kernel, per-phase cycle counters."""
import sys
import numpy as np
sys.path.insert(0, ".")
from genomad_amd import synthetic, _lib  # noqa: E402
from genomad_amd.engine import NNEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
eng = NNEngine(0, synthetic.synth_weights(), chunk=4096)
wins = synthetic.synth_windows(0, n)
taps = ("m_a", "m_b", "yp_a", "yp_b", "feat")
ref, rt = eng.debug_forward(wins, "f32", taps=taps)
x3, xt = eng.debug_forward(wins, "f16x3", taps=taps)
tc, tt = eng.debug_forward(wins, "f16x3tc", taps=taps)
for k in taps:
    print(f"{k:5s} max|tc - f32| {np.abs(tt[k] - rt[k]).max():.3e}   max|x3 - f32| {np.abs(xt[k] - rt[k]).max():.3e}   tc == x3 bitwise: {np.array_equal(tt[k], xt[k])}"
          f"   nan: {int(np.isnan(tt[k]).sum())}")
print(f"scores max|tc - f32| {np.abs(tc - ref).max():.3e}   max|x3 - f32| {np.abs(x3 - ref).max():.3e}")
if len(sys.argv) > 2:
    N = 16384
    b, s = eng.alloc(N * 6000), eng.alloc(N * 12)
    eng.synth_windows_dev(0, N, b.ptr)
    for prec in ("f16x3", "f16x3tc"):
        eng.classify_dev(b.ptr, N, s.ptr, prec)
        eng.sync()
        eng.profile_enable(True)
        eng.profile_reset()
        for _ in range(3):
            eng.classify_dev(b.ptr, N, s.ptr, prec)
        eng.sync()
        ms, l = eng.profile_get(_lib.K_FUSED)
        eng.profile_enable(False)
        print(f"{prec}: fused front end {ms / l:.3f} ms per 4096 windows")
    a = s.download((N, 3), np.float32)
    eng.classify_dev(b.ptr, N, s.ptr, "f32")
    eng.sync()
    print(f"f16x3tc vs exact f32 over {N} windows: max |dscore| {np.abs(a - s.download((N, 3), np.float32)).max():.3e}")
if len(sys.argv) > 2:
    import ctypes as C
    names = ["conv2 loop (b0..b7)", "conv2 epilogue", "wait B1", "w_v A + pool", "conv3 loop", "conv3 epilogue", "wait B0", "w_v B + pool (+loop top)",
             "h conv2 phase: own work between barriers", "h conv2 phase: waiting at b0..b7", "h waiting at B1", "h waiting at B0",
             "h V3 chunks 0,1 (beside w_v A)", "h conv3 phase: own work between barriers", "h conv3 phase: waiting at b'0..b'7", "h V2 chunks 0,1 (beside w_v B)"]
    _lib.check(eng.lib.gnn_phase_cycles(eng.ctx, 1, None))
    eng.classify_dev(b.ptr, 4096, s.ptr, "f16x3tc")
    eng.sync()
    out = (C.c_uint64 * 16)()
    _lib.check(eng.lib.gnn_phase_cycles(eng.ctx, 0, out))
    per = [v / (4096 * 63) for v in out]
    print(f"f16x3tc cycles per 96-row step (instrumented build): matrix wave total {sum(per[:8]):.0f}, helper total {sum(per[8:]):.0f}")
    for nm, v in zip(names, per):
        print(f"    {nm:50s} {v:8.0f}")
