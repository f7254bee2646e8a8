#!/bin/bash
# Builds the three libraries scripts/prow_race_demo.py and scripts/async_hunt.sh compare (run in the build container, where the
# git history is; build_variants/ travels to the GPU box with the snapshot):
#   build_variants/lib_prowold.so    round-2 f16c6 kernel (gnn_fused_c6.hip of commit 91789c4) in this round's library
#   build_variants/lib_race_old.so   the same with helper wave 5 delayed at the top of every 8th step (-DGNN_RACE_DELAY=4)
#   build_variants/lib_race_new.so   this round's kernel with the same delay
set -euo pipefail
cd "$(dirname "$0")/.."
R02=${R02_COMMIT:-91789c4}
T=$(mktemp -d)
git show $R02:genomad_amd/csrc/gnn_fused_c6.hip > $T/gnn_fused_c6_old.hip
python - "$T" <<'PY'
import sys
t = sys.argv[1]
s = open(f"{t}/gnn_fused_c6_old.hip").read()
old = """            if (ht < PROW_N) {
                const int t = t0 + FT6 - CARRY + ht;
                prow[ht] = prow_make(nlo, nhi, t);"""
assert old in s
s = s.replace(old, """#ifdef GNN_RACE_DELAY
            if (wave == 5 && (step % 8) == 3)
                for (int i = 0; i < GNN_RACE_DELAY; ++i) __builtin_amdgcn_s_sleep(127);
#endif
""" + old)
open(f"{t}/gnn_fused_c6_old_delay.hip", "w").write(s)
PY
bash genomad_amd/csrc/build.sh > /dev/null
cd genomad_amd/csrc
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I."
$HIPCC -c $T/gnn_fused_c6_old.hip -o $T/old.o
$HIPCC -DGNN_RACE_DELAY=4 -c $T/gnn_fused_c6_old_delay.hip -o $T/race_old.o
$HIPCC -DGNN_RACE_DELAY=4 -c gnn_fused_c6.hip -o $T/race_new.o
mkdir -p ../../build_variants
for v in prowold:old race_old:race_old race_new:race_new; do
  name=${v%%:*}; obj=${v##*:}
  objs=""
  for f in gnn_api gnn_encode gnn_front_f32 gnn_backend gnn_fused gnn_fused_c8 gnn_fused_c6 gnn_fused_x3 gnn_probe gnn_consumers gnn_fasta gnn_comm gnn_contigs; do
    if [ $f = gnn_fused_c6 ]; then objs="$objs $T/$obj.o"; else objs="$objs obj/$f.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_variants/lib_$name.so $objs -ldl
  echo build_variants/lib_$name.so
done
