#!/bin/bash
# Round-4 evidence run on the GPU box (via gpurun): GPU suite with durations, the default bench line, the reference's call shape
# the same with rocm-smi power samples, (128 windows per call) with and without the time split, a real FASTA through main(), a reduced metagenome run.
#   scripts/gpu_evidence_r04.sh [tag]   -> gpurun_out/<tag>/*
set -u
TAG=${1:-r04e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
(time python -m pytest tests -m gpu -x -q --durations=8) > $OUT/pytest_gpu.txt 2>&1
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --power --steps 48 --cpu-sample 0 --check none > $OUT/bench_power.json 2> $OUT/bench_power.err
python scripts/batch128_bench.py > $OUT/batch128_time_split.txt 2>&1
GNN_NO_TIME_SPLIT=1 python scripts/batch128_bench.py > $OUT/batch128_one_workgroup_per_window.txt 2>&1
python scripts/real_input_bench.py 600 > $OUT/real_input.txt 2>&1
python bench.py --workload metagenome --gbp-total 6 --cpu-sample 0 > $OUT/bench_metagenome_6gbp.json 2> $OUT/bench_metagenome.err
tail -3 $OUT/pytest_gpu.txt; python -c "
import json; o=json.load(open('$OUT/bench_default.json')); print(o['value'], o['roofline']['frac'], o['roofline']['avg_launch_ms'], o['parity']['max_abs_dscore_all'], o['max_abs_dscore'])"
grep "batch   128" $OUT/batch128_*.txt; tail -4 $OUT/real_input.txt; python -c "
import json; o=json.load(open('$OUT/bench_metagenome_6gbp.json')); print(o['value'], o['gbp_per_s'])"
