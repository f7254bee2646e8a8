#!/bin/bash
# Round 5 final evidence run: GPU suite, smoke(), default bench (optionally with power samples), the metagenome at the full 60 Gbp of
# BASELINE configs[4] with roofline + cpu_baseline.          scripts/gpu_final_r05.sh [tag] [--power] [--metagenome]
set -u
TAG=${1:-r05final}
O=gpurun_out/$TAG
mkdir -p $O
( time python -m pytest tests -m gpu -x -q -s ) > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed|error|parity sentinel:|config 3,|seed-43|gnn_classify of 128" $O/pytest_gpu.txt | tail -12
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
PW=""; [[ " $* " == *" --power "* ]] && PW="--power"
python bench.py $PW > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.err
python -c "
import json; o = json.load(open('$O/bench_default.json'))
print('value', o['value'], 'frac', o['roofline']['frac'], 'launch ms', o['roofline']['avg_launch_ms'], 'parity', o.get('parity', {}).get('max_abs_dscore_all'), 'golden', o.get('max_abs_dscore'), 'encoder', o.get('encoder', {}).get('frac'), 'traffic x', o['roofline'].get('traffic_over_algorithmic'), 'power', o.get('power', {}).get('mean_watts_rank0'), 'cpu', o['cpu_baseline']['value'])"
if [[ " $* " == *" --metagenome "* ]]; then
python bench.py --workload metagenome --gbp-total 60 --cpu-sample 256 > $O/bench_metagenome_60gbp.json 2> $O/bench_metagenome.err
python -c "
import json; o = json.load(open('$O/bench_metagenome_60gbp.json'))
print('metagenome', o['value'], o['seconds'], 's frac', o.get('roofline', {}).get('frac'), 'cpu', o.get('cpu_baseline', {}).get('value'))"
fi
