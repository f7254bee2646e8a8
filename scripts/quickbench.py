import sys, time, numpy as np
sys.path.insert(0, '.')
from genomad_amd import synthetic, _lib
from genomad_amd.engine import NNEngine
eng = NNEngine(0, synthetic.synth_weights(), chunk=2048)
print(eng.device_info())
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
bases = eng.alloc(n*6000); scores = eng.alloc(n*12)
eng.synth_windows_dev(0, n, bases.ptr); eng.sync()
for prec in sys.argv[2:] or ['bf16x3','bf16']:
    eng.classify_dev(bases.ptr, n, scores.ptr, prec); eng.sync()
    eng.profile_enable(True); eng.profile_reset()
    t=time.time(); eng.classify_dev(bases.ptr, n, scores.ptr, prec); eng.sync(); dt=time.time()-t
    fms, fl = eng.profile_get(_lib.K_FUSED); bms, bl = eng.profile_get(_lib.K_BACKEND); f32ms,_ = eng.profile_get(_lib.K_F32_FRONT)
    print(f"{prec}: {n} windows in {dt*1e3:.1f} ms = {n/dt:.0f} win/s; fused {fms:.1f} ms ({fl}), f32front {f32ms:.1f}, backend {bms:.1f} ms; useful TF/s {n*2.763e9/dt/1e12:.1f}")
    eng.profile_enable(False)

import ctypes as C
names = ["w_v+pool A","conv2 loop","wait B1","conv2 epi+B2","conv3 loop","wait B3","conv3 epi+B4","w_v+pool B+B0","helper m-partials","helper gather (late part)"]
for prec in sys.argv[2:] or ['bf16x3']:
    if prec == 'f32': continue
    _lib.check(eng.lib.gnn_phase_cycles(eng.ctx, 1, None))
    eng.classify_dev(bases.ptr, n, scores.ptr, prec); eng.sync()
    out = (C.c_uint64*16)()
    _lib.check(eng.lib.gnn_phase_cycles(eng.ctx, 0, out))
    tot = sum(out)
    print(prec, "phase cycles per window-step (wave0):", {nm: round(v/n/47) for nm, v in zip(names, out)}, "total/step", round(tot/n/47))
