#!/bin/bash
# Round 6: the GPU suite alone (no -x: every failure in one run).
set -u
out=gpurun_out/r06_${1:-tests}; mkdir -p $out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1
tail -40 $out/pytest_gpu.txt
