#!/bin/bash
# Round 6 profile run on the GPU box (via gpurun): rocprofv3 kernel trace + stats of the default bench command, the PMC passes of the
# dominant kernel (each in its own run, kernel trace only - never combined with other trace domains), the encoder's WRITE_SIZE /
# FETCH_SIZE passes, and profiles/hbm_traffic.json regenerated from the passes just taken (scripts/hbm_traffic_json.py).
#   scripts/gpu_profile_r06.sh [tag]      -> gpurun_out/<tag>/rocprof/*
set -u
ROOT=$(pwd)
TAG=${1:-r06}
OUT=$ROOT/gpurun_out/$TAG/rocprof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
summarise() {   # $1 counter csv, $2 kernel trace csv -> per-kernel mean duration and mean counter value per dispatch
python - "$1" "$2" <<'PY'
import csv, sys, collections
def tiny(r):   # one-workgroup dispatches of a fused kernel = the calibration windows of gnn_load_weights, not the workload
    if 'fused_front' not in r['Kernel_Name']: return False
    for key in ('Grid_Size', 'Grid_Size_X'):
        if r.get(key) not in (None, ''): return int(r[key]) <= 512
    return False
dur = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[2])):
    if tiny(r): continue
    k = r['Kernel_Name'][:56]
    dur[k][0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6; dur[k][1] += 1
for k, (v, n) in sorted(dur.items()):
    print(f"{k:56s} mean duration {v / n:.4f} ms (n={n})")
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if tiny(r): continue
    k = (r['Kernel_Name'][:56], r['Counter_Name'])
    acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k:56s} {c:36s} mean/dispatch {v / n:.6g}  (n={n})")
PY
}
# 1. kernel trace + stats of the bench command (back-end overlap is off by default since round 4)
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- \
  python $ROOT/bench.py --steps 8 --warmup 1 --cpu-sample 0 --check none --no-extras ${BENCH_ARGS:-} > $OUT/kt.log 2>&1
cp $(find /tmp/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv 2>/dev/null
# 2. PMC passes of the classification (FETCH_SIZE and WRITE_SIZE each alone)
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TA_TA_BUSY_sum TD_TD_BUSY_sum SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$i -- \
    python $ROOT/bench.py --steps 1 --warmup 1 --windows-per-step 4096 --cpu-sample 0 --check none --no-encoder --no-extras ${BENCH_ARGS:-} > $OUT/pmc_$i.log 2>&1
  summarise "$(find /tmp/pmc_$i -name '*counter_collection.csv' | head -1)" "$(find /tmp/pmc_$i -name '*kernel_trace.csv' | head -1)" > $OUT/pmc_$i.txt
done
# 3. the stand-alone encoder: WRITE_SIZE and FETCH_SIZE of onehot_kernel per output dtype
for d in u8 bf16 f32; do
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/enc
    timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/enc -- \
      python $ROOT/bench.py --kernel encoder --onehot-dtype $d --steps 4 --warmup 1 > $OUT/encoder_${d}_$c.log 2>&1
    summarise "$(find /tmp/enc -name '*counter_collection.csv' | head -1)" "$(find /tmp/enc -name '*kernel_trace.csv' | head -1)" | grep -i onehot > $OUT/encoder_${d}_$c.txt
  done
done
cd $ROOT
# 4. FETCH_SIZE against a known byte count for the row-gather access form (MI355X_MICROARCH.md: "calibrate on a known byte count in your
# own access pattern"): one launch per access form over a 128 MiB (Infinity-Cache resident) and a 137 GB table
if [ -x build_variants/probe_gather_big ]; then
  cd /tmp
  for k in 9 14; do
    rm -rf /tmp/cal
    timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/cal -- $ROOT/build_variants/probe_gather_big calib $k > $OUT/fetch_calibration_$k.log 2>&1
    summarise "$(find /tmp/cal -name '*counter_collection.csv' | head -1)" "$(find /tmp/cal -name '*kernel_trace.csv' | head -1)" > $OUT/fetch_calibration_$k.txt
  done
  cd $ROOT
fi
KIND=${PROFILE_KIND:-tk}
python scripts/hbm_traffic_json.py --$KIND-dir $OUT --out $OUT/hbm_traffic.json > $OUT/hbm_traffic_json.log 2>&1
ls -la $OUT
cat $OUT/pmc_1.txt | grep fused; cat $OUT/pmc_2.txt | grep fused; cat $OUT/encoder_*_WRITE_SIZE.txt
