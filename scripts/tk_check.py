"""First-light / regression check of the k-mer-table front end (GNN_PREC_F16X3TK) on the GPU box: table build time, intermediates and
scores against the exact-f32 path and the default arithmetic on clean windows and on windows with N runs / padding / lower case,
padding skip and time split bit-identity, timing of the fused kernel, per-phase cycle counters."""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from genomad_amd import synthetic, _lib  # noqa: E402
from genomad_amd.engine import NNEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
eng = NNEngine(0, synthetic.synth_weights(), chunk=4096)
print("device memory free / total GB:", [round(v / 1e9, 1) for v in eng.mem_info()])
t0 = time.time()
ok = eng.build_kmer_tables()
eng.sync()
print(f"k-mer tables built: {ok} in {time.time() - t0:.2f} s; free now {eng.mem_info()[0] / 1e9:.1f} GB")
if not ok:
    sys.exit(1)
wins = synthetic.synth_windows(0, n)
rng = np.random.default_rng(5)
# dirty windows: N runs, a padded tail, IUPAC codes, lower case, N at the very start / end
dirty = synthetic.synth_windows(100, 8).copy()
dirty[0, 1000:1003] = ord("N")
dirty[0, 2000:2100] = ord("N")
dirty[1, 3500:] = ord("N")
dirty[2, :7] = ord("N")
dirty[2, 5990:] = ord("N")
dirty[3, rng.integers(0, 6000, 40)] = ord("R")
dirty[4, 2500:2600] += 32                      # lower case
dirty[5, :] = ord("N")
dirty[6, 100:] = ord("N")
dirty[7, 5996] = ord("N")
allw = np.concatenate([wins, dirty])
taps = ("m_a", "m_b", "yp_a", "yp_b", "feat")
ref, rt = eng.debug_forward(allw, "f32", taps=taps)
tc, tt = eng.debug_forward(allw, "f16x3tc", taps=taps)
tk, kt = eng.debug_forward(allw, "f16x3tk", taps=taps)
for name, sl in (("clean", slice(0, n)), ("dirty", slice(n, n + 8))):
    for k in taps:
        print(f"{name} {k:5s} max|tk - f32| {np.abs(kt[k][sl] - rt[k][sl]).max():.3e}   max|tc - f32| {np.abs(tt[k][sl] - rt[k][sl]).max():.3e}"
              f"   nan: {int(np.isnan(kt[k][sl]).sum())}")
    print(f"{name} scores max|tk - f32| {np.abs(tk[sl] - ref[sl]).max():.3e}   max|tc - f32| {np.abs(tc[sl] - ref[sl]).max():.3e}")
for i in range(8):
    print(f"  dirty window {i}: max|tk - f32| scores {np.abs(tk[n + i] - ref[n + i]).max():.3e}  yp_b {np.abs(kt['yp_b'][n + i] - rt['yp_b'][n + i]).max():.3e}"
          f"  m_a {np.abs(kt['m_a'][n + i] - rt['m_a'][n + i]).max():.3e}")
# padding skip and time split: bit-identical
for name, fn in (("pad skip", eng.lib.gnn_debug_set_pad_skip), ("time split", eng.lib.gnn_debug_set_time_split)):
    _lib.check(fn(eng.ctx, 0))
    off, ot = eng.debug_forward(allw, "f16x3tk", taps=taps)
    _lib.check(fn(eng.ctx, 1))
    print(f"{name} off == on bitwise: scores {np.array_equal(off, tk)}  " + "  ".join(f"{k} {np.array_equal(ot[k], kt[k])}" for k in taps))
if len(sys.argv) > 2:
    N = 16384
    b, s = eng.alloc(N * 6000), eng.alloc(N * 12)
    eng.synth_windows_dev(0, N, b.ptr)
    for prec in ("f16x3tc", "f16x3tk"):
        eng.classify_dev(b.ptr, N, s.ptr, prec)
        eng.sync()
        eng.profile_enable(True)
        eng.profile_reset()
        for _ in range(3):
            eng.classify_dev(b.ptr, N, s.ptr, prec)
        eng.sync()
        ms, l = eng.profile_get(_lib.K_FUSED)
        eng.profile_enable(False)
        print(f"{prec}: fused front end {ms / l:.3f} ms per 4096 windows = {4096 / (ms / l) * 1e3:.0f} windows/s")
    a = s.download((N, 3), np.float32)
    eng.classify_dev(b.ptr, N, s.ptr, "f32")
    eng.sync()
    print(f"f16x3tk vs exact f32 over {N} windows: max |dscore| {np.abs(a - s.download((N, 3), np.float32)).max():.3e}")
    import ctypes as C
    names = ["conv3 loop (c0..c7)", "conv3 epilogue", "wait E", "w_v B + pool", "V3 chunk 1 + table pool", "loop top", "prologue / 63", "-",
             "h waiting at the barriers (all)", "h c0 .. c1", "h c1 .. c5", "h c5 .. c6", "h c6 .. c7: x2 rows -> LDS, head A's table read",
             "h c7 .. E: V3 chunk 0, pair weights", "h pair products of the step before .. c0", "h E .. the burst is out: this entry, next rows' indices, 13 requests"]
    _lib.check(eng.lib.gnn_phase_cycles(eng.ctx, 1, None))
    eng.classify_dev(b.ptr, 4096, s.ptr, "f16x3tk")
    eng.sync()
    out = (C.c_uint64 * 16)()
    _lib.check(eng.lib.gnn_phase_cycles(eng.ctx, 0, out))
    per = [v / (4096 * 63) for v in out]
    print(f"f16x3tk cycles per 96-row step (instrumented build): matrix wave total {sum(per[:8]):.0f}, helper total {sum(per[8:]):.0f}")
    for nm, v in zip(names, per):
        print(f"    {nm:50s} {v:8.0f}")
