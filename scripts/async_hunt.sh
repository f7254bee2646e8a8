#!/bin/bash
# Fresh-process hunt for the asynchronous-path mismatch (profiles/history/r02c6_async_flake.md): scripts/async_hunt.py once per
# process, N processes per setting, failures per setting and the first diagnostics.
# Usage (GPU box): bash scripts/async_hunt.sh [N per setting, default 60] [setting ... | auto] > gpurun_out/async_hunt.txt
# "auto": the two orderings with the round-2 kernel first; the bisecting settings only if one of them failed on this box
# (the mismatch was seen on one box of seven: every gpurun call is another box, so every call samples one).
set -u
cd "$(dirname "$0")/.."
N=${1:-60}
shift || true
ONLY="$*"
echo "box: $(hostname) $(rocm-smi --showuniqueid 2>/dev/null | grep -i 'unique' | head -1 | tr -s ' ')"
TOTAL_FAIL=0
arm() {   # name, env assignments...
  local name=$1; shift
  if [ -n "$ONLY" ] && [ "$ONLY" != auto ] && ! grep -qw "$name" <<<"$ONLY"; then return; fi
  local fail=0 t0=$(date +%s)
  for i in $(seq 1 "$N"); do
    out=$(env "$@" timeout 120 python scripts/async_hunt.py 2 ${HUNT_PREC:-f16c6} 2>&1 | tail -1)
    if [[ "$out" != OK* ]]; then
      fail=$((fail + 1))
      [ "$fail" -le 4 ] && echo "  [$name #$i] ${out:0:1500}"
    fi
  done
  echo "$name: $fail / $N failed ($(( $(date +%s) - t0 )) s)"
  TOTAL_FAIL=$((TOTAL_FAIL + fail))
}
arm old_order        GNN_ASYNC_EVENT_WAIT=1 GENOMAD_AMD_LIB=build_variants/lib_prowold.so
arm old_order_newk   GNN_ASYNC_EVENT_WAIT=1
arm default          GNN_DUMMY=0
HUNT_PREC=f16x3 arm default_f16x3 GNN_DUMMY=0          # the default arithmetic: the streaming kernel of gnn_fused_x3.hip
if [ "$ONLY" = auto ] && [ "$TOTAL_FAIL" -eq 0 ]; then echo "auto: nothing failed on this box, bisecting settings skipped"; exit 0; fi
arm new_order_oldk   GENOMAD_AMD_LIB=build_variants/lib_prowold.so
arm poison           GNN_DEBUG_POISON=1
arm old_no_overlap   GNN_ASYNC_EVENT_WAIT=1 GENOMAD_AMD_LIB=build_variants/lib_prowold.so GNN_NO_BACKEND_OVERLAP=1
arm old_serial_k     GNN_ASYNC_EVENT_WAIT=1 GENOMAD_AMD_LIB=build_variants/lib_prowold.so AMD_SERIALIZE_KERNEL=3
arm old_one_queue    GNN_ASYNC_EVENT_WAIT=1 GENOMAD_AMD_LIB=build_variants/lib_prowold.so GPU_MAX_HW_QUEUES=1
arm old_no_sdma      GNN_ASYNC_EVENT_WAIT=1 GENOMAD_AMD_LIB=build_variants/lib_prowold.so HSA_ENABLE_SDMA=0
arm old_no_padskip   GNN_ASYNC_EVENT_WAIT=1 GENOMAD_AMD_LIB=build_variants/lib_prowold.so GNN_NO_PAD_SKIP=1
