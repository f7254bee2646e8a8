#!/bin/bash
# Round 6 evidence run: GPU suite, first-light checks of both Toom-Cook kernels, the default bench line (f16x3tk when the device holds the
# k-mer tables) and the same command with --no-kmer-tables (f16x3tc), rocprofv3 kernel trace + PMC passes (scripts/gpu_profile_r06.sh).
set -u
tag=${1:-final}
out=gpurun_out/r06_$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -rs > $out/pytest_gpu.txt 2>&1
timeout 300 python scripts/tk_check.py 32 t > $out/tk_check.txt 2>&1
timeout 300 python scripts/tc_check.py 32 t > $out/tc_check.txt 2>&1
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
timeout 900 python bench.py --no-kmer-tables --no-extras > $out/bench_no_kmer_tables.json 2> $out/bench_no_kmer_tables.err
timeout 900 python bench.py --workload metagenome --gbp-total 60 > $out/bench_metagenome_60gbp.json 2> $out/bench_metagenome_60gbp.err
timeout 300 python scripts/tk_nrich.py > $out/nrich.txt 2>&1
timeout 2400 bash scripts/gpu_profile_r06.sh r06_$tag > $out/profile.log 2>&1
tail -6 $out/pytest_gpu.txt; tail -22 $out/tk_check.txt; cut -c1-300 $out/bench_default.json; echo; cut -c1-200 $out/bench_no_kmer_tables.json; echo; cut -c1-260 $out/bench_metagenome_60gbp.json; echo; tail -4 $out/nrich.txt; tail -14 $out/profile.log; cat $out/rocprof/fetch_calibration_*.txt
