#!/bin/bash
# Build libgenomad_nn_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.  The flags are fixed here: a measurement
# variant is built by scripts/mkvariant.sh into build_variants/ and never replaces the in-tree library.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result"
STEMS="gnn_api gnn_encode gnn_front_f32 gnn_backend gnn_pack gnn_fused_c6 gnn_fused_x3 gnn_fused_tc gnn_fused_tk gnn_probe gnn_consumers gnn_fasta gnn_comm gnn_contigs"
HDRS="gnn_common.h gnn_fused_common.h gnn_fused_helpers.h gnn_tc_dev.h ../../include/genomad_nn.h"
stale() { [ ! -f "$2" ] && return 0; for d in "$1" $HDRS; do [ "$d" -nt "$2" ] && return 0; done; return 1; }
mkdir -p obj
pids=()
for f in $STEMS; do
  if stale $f.hip obj/$f.o; then
    # gnn_fused_tc: no SLP vectorisation - the helpers' transform runs beside the MFMA stream, where v_pk_*_f32 issue worse than
    # two scalar ops (MI355X_MICROARCH.md, "price of one filler beside MFMAs"; 23.5 vs 24.1 ms per 4096 windows)
    PERFILE=""; { [ $f = gnn_fused_tc ] || [ $f = gnn_fused_tk ]; } && PERFILE="-fno-slp-vectorize"
    $HIPCC $FLAGS $PERFILE -c $f.hip -o obj/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
OBJS=""; for f in $STEMS; do OBJS="$OBJS obj/$f.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libgenomad_nn_hip.so $OBJS -ldl
echo "built $(pwd)/libgenomad_nn_hip.so"
# Test variant (tests/test_gpu_parity.py::test_toomcook_kernel_is_bit_identical_under_delay_injection): the same library with random
# sleeps behind every barrier of the default kernel.  Never loaded by the product; built with the main library so that it travels
# to the GPU box.
for k in gnn_fused_tc gnn_fused_tk; do
  if stale $k.hip obj/${k}_jitter.o; then
    $HIPCC $FLAGS -fno-slp-vectorize -DTC_JITTER -c $k.hip -o obj/${k}_jitter.o
  fi
done
JOBJS=${OBJS/obj\/gnn_fused_tc.o/obj\/gnn_fused_tc_jitter.o}
JOBJS=${JOBJS/obj\/gnn_fused_tk.o/obj\/gnn_fused_tk_jitter.o}
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libgenomad_nn_hip_jitter.so $JOBJS -ldl
echo "built $(pwd)/libgenomad_nn_hip_jitter.so"
