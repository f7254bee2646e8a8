#!/bin/bash
# Round 6 evidence run: GPU suite, default bench line, rocprofv3 kernel trace + PMC passes (scripts/gpu_profile_r06.sh), per-phase cycles.
set -u
tag=${1:-final}
out=gpurun_out/r06_$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1
timeout 300 python scripts/tc_check.py 32 t > $out/tc_check.txt 2>&1
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
timeout 2400 bash scripts/gpu_profile_r06.sh r06_$tag > $out/profile.log 2>&1
tail -4 $out/pytest_gpu.txt; tail -18 $out/tc_check.txt; cut -c1-300 $out/bench_default.json; tail -12 $out/profile.log
