#!/bin/bash
# Build libgenomad_nn_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result"
mkdir -p obj
pids=()
for f in gnn_api gnn_encode gnn_front_f32 gnn_backend gnn_fused gnn_fused_c8 gnn_fused_c6 gnn_fused_x3 gnn_fused_tc gnn_probe gnn_consumers gnn_fasta gnn_comm gnn_contigs; do
  if [ ! -f obj/$f.o ] || [ $f.hip -nt obj/$f.o ] || [ gnn_common.h -nt obj/$f.o ] || [ gnn_fused_common.h -nt obj/$f.o ] || [ gnn_fused_helpers.h -nt obj/$f.o ] || [ ../../include/genomad_nn.h -nt obj/$f.o ]; then
    # gnn_fused_tc: no SLP vectorisation - the helpers' transform runs beside the MFMA stream, where v_pk_*_f32 issue worse than
    # two scalar ops (MI355X_MICROARCH.md, "price of one filler beside MFMAs"; 23.5 vs 24.1 ms per 4096 windows)
    PERFILE=""; [ $f = gnn_fused_tc ] && PERFILE="-fno-slp-vectorize"
    $HIPCC $FLAGS $PERFILE ${EXTRA_FLAGS:-} -c $f.hip -o obj/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libgenomad_nn_hip.so obj/gnn_api.o obj/gnn_encode.o obj/gnn_front_f32.o obj/gnn_backend.o obj/gnn_fused.o obj/gnn_fused_c8.o obj/gnn_fused_c6.o obj/gnn_fused_x3.o obj/gnn_fused_tc.o obj/gnn_probe.o obj/gnn_consumers.o obj/gnn_fasta.o obj/gnn_comm.o obj/gnn_contigs.o -ldl
echo "built $(pwd)/libgenomad_nn_hip.so"
# Test variant (tests/test_gpu_parity.py::test_toomcook_kernel_is_bit_identical_under_delay_injection): the same library with random
# sleeps behind every barrier of the default kernel.  Never loaded by the product; built with the main library so that it travels
# to the GPU box.
stale=0
for o in obj/gnn_api.o obj/gnn_encode.o obj/gnn_front_f32.o obj/gnn_backend.o obj/gnn_fused.o obj/gnn_fused_c8.o obj/gnn_fused_c6.o obj/gnn_fused_x3.o obj/gnn_fused_tc.o obj/gnn_probe.o obj/gnn_consumers.o obj/gnn_fasta.o obj/gnn_comm.o obj/gnn_contigs.o; do
  [ $o -nt libgenomad_nn_hip_jitter.so ] && stale=1
done
if [ ! -f libgenomad_nn_hip_jitter.so ] || [ $stale = 1 ]; then
  $HIPCC $FLAGS -fno-slp-vectorize -DTC_JITTER -c gnn_fused_tc.hip -o obj/gnn_fused_tc_jitter.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o libgenomad_nn_hip_jitter.so obj/gnn_api.o obj/gnn_encode.o obj/gnn_front_f32.o obj/gnn_backend.o obj/gnn_fused.o obj/gnn_fused_c8.o obj/gnn_fused_c6.o obj/gnn_fused_x3.o obj/gnn_fused_tc_jitter.o obj/gnn_probe.o obj/gnn_consumers.o obj/gnn_fasta.o obj/gnn_comm.o obj/gnn_contigs.o -ldl
  echo "built $(pwd)/libgenomad_nn_hip_jitter.so"
fi
