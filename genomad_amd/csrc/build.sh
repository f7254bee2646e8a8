#!/bin/bash
# Build libgenomad_nn_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
#   GNN_EXPERIMENTAL=1 build.sh    links the experimental f16c8 kernel (gnn_fused_c8.hip) instead of its stub: GNN_PREC_F16C8 fails
#                                  the score tolerance on 10^6 windows, is frozen, and is not part of the default library.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result"
C8=gnn_fused_c8_stub; [ "${GNN_EXPERIMENTAL:-0}" = 1 ] && C8=gnn_fused_c8
STEMS="gnn_api gnn_encode gnn_front_f32 gnn_backend gnn_fused $C8 gnn_fused_c6 gnn_fused_x3 gnn_fused_tc gnn_probe gnn_consumers gnn_fasta gnn_comm gnn_contigs"
mkdir -p obj
pids=()
for f in $STEMS; do
  if [ ! -f obj/$f.o ] || [ $f.hip -nt obj/$f.o ] || [ gnn_common.h -nt obj/$f.o ] || [ gnn_fused_common.h -nt obj/$f.o ] || [ gnn_fused_helpers.h -nt obj/$f.o ] || [ ../../include/genomad_nn.h -nt obj/$f.o ]; then
    # gnn_fused_tc: no SLP vectorisation - the helpers' transform runs beside the MFMA stream, where v_pk_*_f32 issue worse than
    # two scalar ops (MI355X_MICROARCH.md, "price of one filler beside MFMAs"; 23.5 vs 24.1 ms per 4096 windows)
    PERFILE=""; [ $f = gnn_fused_tc ] && PERFILE="-fno-slp-vectorize"
    $HIPCC $FLAGS $PERFILE ${EXTRA_FLAGS:-} -c $f.hip -o obj/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
OBJS=""; for f in $STEMS; do OBJS="$OBJS obj/$f.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libgenomad_nn_hip.so $OBJS -ldl
echo "built $(pwd)/libgenomad_nn_hip.so ($C8)"
# Test variant (tests/test_gpu_parity.py::test_toomcook_kernel_is_bit_identical_under_delay_injection): the same library with random
# sleeps behind every barrier of the default kernel.  Never loaded by the product; built with the main library so that it travels
# to the GPU box.
if [ ! -f obj/gnn_fused_tc_jitter.o ] || [ gnn_fused_tc.hip -nt obj/gnn_fused_tc_jitter.o ] || [ gnn_common.h -nt obj/gnn_fused_tc_jitter.o ] || [ gnn_fused_common.h -nt obj/gnn_fused_tc_jitter.o ] || [ gnn_fused_helpers.h -nt obj/gnn_fused_tc_jitter.o ] || [ ../../include/genomad_nn.h -nt obj/gnn_fused_tc_jitter.o ]; then
  $HIPCC $FLAGS -fno-slp-vectorize -DTC_JITTER -c gnn_fused_tc.hip -o obj/gnn_fused_tc_jitter.o
fi
# always relinked (one second): the same objects as the main library, whichever f16c8 file that is
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libgenomad_nn_hip_jitter.so ${OBJS/obj\/gnn_fused_tc.o/obj\/gnn_fused_tc_jitter.o} -ldl
echo "built $(pwd)/libgenomad_nn_hip_jitter.so"
