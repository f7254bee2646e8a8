#!/bin/bash
# Round-2 evidence run on the GPU box (via gpurun): bench lines, rocprofv3 kernel trace, PMC passes.
#   scripts/gpu_profile.sh [tag]      -> gpurun_out/<tag>/*
set -u
ROOT=$(pwd)
TAG=${1:-r02}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -z "${ONLY_PROFILE:-}" ]; then
# 1. un-profiled bench lines
timeout 600 python $ROOT/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 300 python $ROOT/bench.py --precision f16c8 --steps 16 --cpu-sample 0 > $OUT/bench_f16c8.json 2>> $OUT/bench_default.err
timeout 300 python $ROOT/bench.py --precision bf16x3 --steps 16 --cpu-sample 0 > $OUT/bench_bf16x3.json 2>> $OUT/bench_default.err
timeout 300 python $ROOT/bench.py --precision f16x3 --steps 16 --cpu-sample 0 > $OUT/bench_f16x3.json 2>> $OUT/bench_default.err
timeout 300 python $ROOT/bench.py --steps 8 --force-dist --cpu-sample 0 > $OUT/bench_force_dist.json 2> $OUT/bench_force_dist.err
timeout 300 env WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 \
  python $ROOT/bench.py --steps 8 --scaling weak --force-dist --cpu-sample 0 > $OUT/bench_weak.json 2>> $OUT/bench_force_dist.err
timeout 600 python $ROOT/bench.py --workload metagenome --gbp-total 3 --cpu-sample 0 > $OUT/bench_metagenome_3gbp.json 2> $OUT/bench_metagenome.err
timeout 900 python $ROOT/bench.py --workload metagenome --gbp-total 60 --cpu-sample 0 > $OUT/bench_metagenome_60gbp.json 2>> $OUT/bench_metagenome.err
# 1b. accuracy tails of the fused arithmetic options against the exact f32 device path
(cd $ROOT && timeout 200 python scripts/seed_check.py 10000 42 43; timeout 300 python scripts/seed_check.py 100000 42 43; timeout 400 python scripts/seed_check.py 1048576 42) > $OUT/tails.txt 2>&1
# 1c. what the launch time is made of: ablation builds of the default kernel (scripts/mkvariant.sh c6<name> gnn_fused_c6 -DGNN_ABL_...)
(cd $ROOT && python scripts/ablate_c6.py; for v in $(ls build_variants/lib_c6*.so 2>/dev/null); do GENOMAD_AMD_LIB=$v timeout 120 python scripts/ablate_c6.py; done; python scripts/ablate_c6.py) > $OUT/ablation_c6.txt 2>&1
(cd $ROOT && timeout 300 python scripts/c6_check.py 8192) > $OUT/c6_check.txt 2>&1
(cd $ROOT && timeout 300 python scripts/real_input_bench.py 600) > $OUT/real_input.txt 2>&1
fi
# 2. kernel trace + stats of the default command (shorter)
rm -rf /tmp/kt
# (back-end overlap off: concurrent kernels stretch each other's trace durations; the one-window dispatch in the trace is the
#  all-N calibration window of gnn_load_weights)
timeout 600 env GNN_NO_BACKEND_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- \
  python $ROOT/bench.py --steps 8 --warmup 1 --cpu-sample 0 > $OUT/kt.log 2>&1
cp $(find /tmp/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv 2>/dev/null
# 3. PMC passes (own runs, kernel trace only)
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TA_TA_BUSY_sum TD_TD_BUSY_sum SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$i -- \
    python $ROOT/bench.py --steps 1 --warmup 1 --windows-per-step 4096 --cpu-sample 0 > $OUT/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name '*counter_collection.csv' | head -1)
  g=$(find /tmp/pmc_$i -name '*kernel_trace.csv' | head -1)
  python - "$f" "$g" > $OUT/pmc_$i.txt <<'PY'
import csv, sys, collections
def tiny(r):   # one-workgroup dispatches of the fused kernel = the calibration window, not the workload
    if 'fused_front_c6' not in r['Kernel_Name']: return False
    for key in ('Grid_Size', 'Grid_Size_X'):
        if r.get(key) not in (None, ''): return int(r[key]) <= 512
    return False
dur = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[2])):
    if tiny(r): continue   # the all-N calibration window of gnn_load_weights
    k = r['Kernel_Name'][:48]
    dur[k][0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6; dur[k][1] += 1
for k, (v, n) in sorted(dur.items()):
    print(f"{k:48s} mean duration {v / n:.4f} ms (n={n})")
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if tiny(r): continue
    k = (r['Kernel_Name'][:48], r['Counter_Name'])
    acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k:48s} {c:36s} mean/dispatch {v / n:.6g}  (n={n})")
PY
done
ls -la $OUT
