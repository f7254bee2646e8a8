#!/bin/bash
# Round 5 evidence run on the GPU box (via gpurun): GPU suite, default bench (+ encoder block), encoder stand-alone lines,
# reduced metagenome with roofline + cpu_baseline.    scripts/gpu_evidence_r05.sh [tag]  -> gpurun_out/<tag>/*
set -u
TAG=${1:-r05}
OUT=gpurun_out/$TAG
mkdir -p $OUT
( time python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.err
python -c "
import json; o = json.load(open('$OUT/bench_default.json'))
print('value', o['value'], 'frac', o['roofline']['frac'], 'parity', o.get('parity', {}).get('max_abs_dscore_all'), 'encoder', o.get('encoder', {}).get('frac'), 'traffic x', o['roofline'].get('traffic_over_algorithmic'))"
for d in u8 bf16 f32; do python bench.py --kernel encoder --onehot-dtype $d --steps 20 > $OUT/bench_encoder_$d.json 2>> $OUT/bench_encoder.err; done
python bench.py --workload metagenome --gbp-total 6 --cpu-sample 256 > $OUT/bench_metagenome_6gbp.json 2> $OUT/bench_metagenome.err
python -c "
import json; o = json.load(open('$OUT/bench_metagenome_6gbp.json'))
print('metagenome', o['value'], 'frac', o.get('roofline', {}).get('frac'), 'cpu', o.get('cpu_baseline', {}).get('value'))"
