#!/bin/bash
# Round 6, batch 1: head A's y @ w_v as a table lookup - correctness, A/B against the round-5 kernel on one box, the GPU suite, the bench line.
set -u
out=gpurun_out/r06_batch1; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python scripts/tc_check.py 32 t > $out/tc_check.txt 2>&1
timeout 900 python scripts/tc_ab2.py --rounds 2 build_variants/lib_base.so default > $out/ab_wva_table.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -5 $out/tc_check.txt; cat $out/ab_wva_table.txt; tail -5 $out/pytest_gpu.txt; cut -c1-400 $out/bench_default.json
