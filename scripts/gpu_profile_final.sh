#!/bin/bash
# Short re-profile of the round's final code: rocprofv3 kernel trace + stats of the default bench and ONE PMC pass (matrix-pipe
# busy cycles, clock, LDS conflicts) -> gpurun_out/<tag>/ ; the full set of passes is scripts/gpu_profile_r03.sh.
set -u
ROOT=$(pwd); TAG=${1:-r03l}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt /tmp/pmc_f
timeout 300 env GNN_NO_BACKEND_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- \
  python $ROOT/bench.py --steps 8 --warmup 1 --cpu-sample 0 --check none --fast-mode-steps 0 > $OUT/kt.log 2>&1
cp $(find /tmp/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_f -- \
  python $ROOT/bench.py --steps 1 --warmup 1 --windows-per-step 4096 --cpu-sample 0 --check none --fast-mode-steps 0 > $OUT/pmc.log 2>&1
python - "$(find /tmp/pmc_f -name '*counter_collection.csv' | head -1)" "$(find /tmp/pmc_f -name '*kernel_trace.csv' | head -1)" > $OUT/pmc_mfma.txt <<'PY'
import csv, sys, collections
def tiny(r):
    if 'fused_front' not in r['Kernel_Name']: return False
    for key in ('Grid_Size', 'Grid_Size_X'):
        if r.get(key) not in (None, ''): return int(r[key]) <= 512
    return False
dur = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[2])):
    if tiny(r): continue
    k = r['Kernel_Name'][:56]; dur[k][0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6; dur[k][1] += 1
for k, (v, n) in sorted(dur.items()): print(f"{k:56s} mean duration {v / n:.4f} ms (n={n})")
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if tiny(r): continue
    k = (r['Kernel_Name'][:56], r['Counter_Name']); acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
for (k, c), (v, n) in sorted(acc.items()): print(f"{k:56s} {c:36s} mean/dispatch {v / n:.6g}  (n={n})")
PY
head -4 $OUT/kernel_stats.csv; grep fused_front_x3 $OUT/pmc_mfma.txt
