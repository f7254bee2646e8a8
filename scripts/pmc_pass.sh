#!/bin/bash
# One rocprofv3 PMC pass per argument (a quoted, space-separated counter set) over a short bench run;
# per-kernel means land in gpurun_out/pmc_<i>.txt.  Run on the GPU box: [BENCH_ARGS='--precision f16c6'] [PMC_TAG=x] scripts/pmc_pass.sh "A B" "C D" ...
set -u
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$i -- \
    python $ROOT/bench.py --steps 1 --warmup 1 --windows-per-step 4096 --cpu-sample 0 ${BENCH_ARGS:-} > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name '*counter_collection.csv' | head -1)
  g=$(find /tmp/pmc_$i -name '*kernel_trace.csv' | head -1)
  python - "$f" "$g" > $ROOT/gpurun_out/pmc${PMC_TAG:-}_$i.txt <<'PY'
import csv, sys, collections
def tiny(r):   # one-workgroup dispatches of the fused kernel = the calibration window, not the workload
    if 'fused_front_c6' not in r['Kernel_Name']: return False
    for key in ('Grid_Size', 'Grid_Size_X'):
        if r.get(key) not in (None, ''): return int(r[key]) <= 512
    return False
dur = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[2])):
    if tiny(r): continue   # the all-N calibration window of gnn_load_weights
    k = r['Kernel_Name'][:40]
    dur[k][0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6; dur[k][1] += 1
for k, (v, n) in sorted(dur.items()):
    if 'fused' in k:
        print(f"{k:40s} mean duration {v / n:.3f} ms (n={n})")
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if tiny(r): continue
    k = (r['Kernel_Name'][:40], r['Counter_Name'])
    acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    if 'fused' in k or 'logits' in k:
        print(f"{k:40s} {c:40s} mean/dispatch {v / n:.6g}  (n={n})")
PY
  cat $ROOT/gpurun_out/pmc${PMC_TAG:-}_$i.txt
done
