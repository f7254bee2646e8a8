#!/bin/bash
# Round 5, third GPU call: the energy A/B with the per-phase cycle counters of every build (the C of the time model), and the
# pricing probe of the conv loops at the F(3,6) and F(4,6) geometries (scripts/probe_tc_loop.hip).
set -u
O=gpurun_out/r05c
mkdir -p $O
V=build_variants
timeout 900 python scripts/tc_ab2.py default default,GNN_TC_WLO_MASK=0 $V/lib_tc_alo0.so $V/lib_tc_alo0.so,GNN_TC_WLO_MASK=0 $V/lib_tc_alo5.so,GNN_TC_WLO_MASK=FFE0 \
  $V/lib_tc_wva_drop.so default $V/lib_tcabl_NOTRANSFORM.so $V/lib_tcabl_NOPAIRS.so $V/lib_tcabl_NOGATHER.so $V/lib_tc_helpnone.so $V/lib_tcabl_NOWV.so \
  $V/lib_tcabl_NOCONVMMA.so $V/lib_tcabl_NOWEIGHTS.so $V/lib_tcabl_NOVREAD.so default $V/lib_tcabl_NOEPI.so $V/lib_tcabl_GATHER_ONE.so \
  $V/lib_tc_sleep16.so $V/lib_tc_sleep48.so $V/lib_tc_sleep96.so default > $O/tc_ab2_cycles.txt 2>&1
cut -c1-330 $O/tc_ab2_cycles.txt
for p in probe_tc_f36 probe_tc_f46_ring8 probe_tc_f46_ring6 probe_tc_f36; do echo "== $p"; timeout 120 $V/$p 4096; done > $O/probe_tc_f36_f46.txt 2>&1
cat $O/probe_tc_f36_f46.txt
