#!/bin/bash
# Build an A/B variant of the library: scripts/mkvariant.sh <name> <file-stem> "<extra hipcc flags>"
# -> build_variants/lib_<name>.so (select with GENOMAD_AMD_LIB=...).  build_variants/ is git-ignored.
set -euo pipefail
name=$1; stem=$2; flags=${3:-}
cd "$(dirname "$0")/../genomad_amd/csrc"
mkdir -p ../../build_variants
perfile=""; { [ $stem = gnn_fused_tc ] || [ $stem = gnn_fused_tk ]; } && perfile="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $perfile $flags -c $stem.hip -o /tmp/variant_$name.o
objs=""
for f in gnn_api gnn_encode gnn_front_f32 gnn_backend gnn_pack gnn_fused_c6 gnn_fused_x3 gnn_fused_tc gnn_fused_tk gnn_probe gnn_consumers gnn_fasta gnn_comm gnn_contigs; do
  if [ $f = $stem ]; then objs="$objs /tmp/variant_$name.o"; else objs="$objs obj/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_variants/lib_$name.so $objs -ldl
echo build_variants/lib_$name.so
