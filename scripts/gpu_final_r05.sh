#!/bin/bash
# Round 5 final evidence run: GPU suite, default bench (+ power samples), the metagenome at the full 60 Gbp of BASELINE configs[4]
# with roofline + cpu_baseline, smoke().          scripts/gpu_final_r05.sh [tag]
set -u
TAG=${1:-r05final}
O=gpurun_out/$TAG
mkdir -p $O
( time python -m pytest tests -m gpu -x -q -s ) > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed|error|parity sentinel:|config 3,|seed-43|RCCL:" $O/pytest_gpu.txt | tail -12
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py --power > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.err
python -c "
import json; o = json.load(open('$O/bench_default.json'))
print('value', o['value'], 'frac', o['roofline']['frac'], 'parity', o.get('parity', {}).get('max_abs_dscore_all'), 'golden', o.get('max_abs_dscore'), 'encoder', o.get('encoder', {}).get('frac'), 'traffic x', o['roofline'].get('traffic_over_algorithmic'), 'power', o.get('power'))"
python bench.py --workload metagenome --gbp-total 60 --cpu-sample 256 > $O/bench_metagenome_60gbp.json 2> $O/bench_metagenome.err
python -c "
import json; o = json.load(open('$O/bench_metagenome_60gbp.json'))
print('metagenome', o['value'], o['seconds'], 's frac', o.get('roofline', {}).get('frac'), 'cpu', o.get('cpu_baseline', {}).get('value'))"
