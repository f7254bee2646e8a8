#!/bin/bash
# Round 5, first GPU call: evidence run (tests, bench, encoder, metagenome), launch-tail pricing, energy A/B with power sampling.
set -u
bash scripts/gpu_evidence_r05.sh r05a
O=gpurun_out/r05a
timeout 300 python scripts/launch_tail.py > $O/launch_tail.txt 2>&1
V=build_variants
timeout 1500 python scripts/tc_ab2.py default default,GNN_TC_WLO_MASK=0 default,GNN_TC_WLO_MASK=FFE0 $V/lib_tc_alo0.so $V/lib_tc_alo5.so $V/lib_tc_alo3.so \
  $V/lib_tc_alo5.so,GNN_TC_WLO_MASK=FFE0 $V/lib_tc_alo0.so,GNN_TC_WLO_MASK=0 $V/lib_tc_wva_drop.so default \
  $V/lib_tcabl_NOTRANSFORM.so $V/lib_tcabl_NOPAIRS.so $V/lib_tcabl_NOGATHER.so $V/lib_tc_helpnone.so $V/lib_tcabl_NOWV.so $V/lib_tcabl_NOCONVMMA.so \
  $V/lib_tcabl_NOWEIGHTS.so $V/lib_tcabl_NOVREAD.so default $V/lib_tcabl_NOEPI.so $V/lib_tcabl_GATHER_ONE.so $V/lib_tc_sleep16.so $V/lib_tc_sleep48.so default \
  > $O/tc_ab2.txt 2>&1
cat $O/launch_tail.txt | tail -30
cat $O/tc_ab2.txt
