#!/bin/bash
# Round 6: A/B of library variants on one box.  usage: gpu_ab_r06.sh <tag> <rounds> spec...   (specs as scripts/tc_ab2.py takes them)
set -u
tag=$1; rounds=$2; shift 2
out=gpurun_out/r06_$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python scripts/tc_ab2.py --rounds $rounds "$@" > $out/ab.txt 2>&1
cat $out/ab.txt
