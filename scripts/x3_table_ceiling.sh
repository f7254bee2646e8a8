#!/bin/bash
# Pricing run for "conv2 as a 9-mer table lookup" on the default arithmetic: what is left on the matrix
# pipe when conv2's MFMAs are gone (ablation build lib_x3noconv2.so: wrong results), what the gather of the table rows
# sustains alone, and what both sustain when they run BESIDE each other (two processes on the one GPU).
cd "$(dirname "$0")/.."
T='import sys,time; sys.path.insert(0,"."); from genomad_amd import synthetic,_lib; from genomad_amd.engine import NNEngine
eng=NNEngine(0,synthetic.synth_weights(),chunk=4096); n=16384; b=eng.alloc(n*6000); s=eng.alloc(n*12); eng.synth_windows_dev(0,n,b.ptr)
secs=float(sys.argv[1]); t0=time.time()
while time.time()-t0<secs:
    eng.classify_dev(b.ptr,n,s.ptr,"f16x3"); eng.sync(); eng.profile_enable(True); eng.profile_reset()
    for _ in range(4): eng.classify_dev(b.ptr,n,s.ptr,"f16x3")
    eng.sync(); ms,l=eng.profile_get(_lib.K_FUSED); eng.profile_enable(False)
    print(f"  fused front end: {ms/l:.3f} ms per 4096 windows", flush=True)'
echo "== default library alone"; python -c "$T" 3
echo "== conv2 MFMAs compiled out, alone"; GENOMAD_AMD_LIB=build_variants/lib_x3noconv2.so python -c "$T" 3
echo "== gather alone"; ./build_variants/probe_gather loop 2
echo "== conv2 MFMAs compiled out BESIDE the gather"
./build_variants/probe_gather loop 14 & GP=$!
sleep 3
GENOMAD_AMD_LIB=build_variants/lib_x3noconv2.so python -c "$T" 6
wait $GP
