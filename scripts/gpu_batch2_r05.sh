#!/bin/bash
# Round 5, second GPU call: the A/B of batch 1 again with the per-phase cycle counters of every build (the C of the time model), then
# the rocprofv3 evidence (kernel trace + PMC passes + encoder passes) of the shipped kernel.
set -u
O=gpurun_out/r05b
mkdir -p $O
V=build_variants
timeout 900 python scripts/tc_ab2.py default default,GNN_TC_WLO_MASK=0 $V/lib_tc_alo0.so $V/lib_tc_alo0.so,GNN_TC_WLO_MASK=0 $V/lib_tc_alo5.so,GNN_TC_WLO_MASK=FFE0 \
  $V/lib_tc_wva_drop.so default $V/lib_tcabl_NOTRANSFORM.so $V/lib_tcabl_NOPAIRS.so $V/lib_tcabl_NOGATHER.so $V/lib_tc_helpnone.so $V/lib_tcabl_NOWV.so \
  $V/lib_tcabl_NOCONVMMA.so $V/lib_tcabl_NOWEIGHTS.so $V/lib_tcabl_NOVREAD.so default $V/lib_tcabl_NOEPI.so $V/lib_tcabl_GATHER_ONE.so \
  $V/lib_tc_sleep16.so $V/lib_tc_sleep48.so $V/lib_tc_sleep96.so default > $O/tc_ab2_cycles.txt 2>&1
cat $O/tc_ab2_cycles.txt
bash scripts/gpu_profile_r05.sh r05b
