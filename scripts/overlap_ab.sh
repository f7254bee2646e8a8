#!/bin/bash
# Back-end overlap policy A/B on one box: the library's default (synchronous calls run their chunks' back ends in order) against
# GNN_BACKEND_OVERLAP=1 (the policy of rounds 2-4), at the bench's default step and at one launch per step.
for w in 65536 16384 65536 16384; do for e in default overlap; do
  if [ $e = overlap ]; then export GNN_BACKEND_OVERLAP=1; else unset GNN_BACKEND_OVERLAP; fi
  s=$((1048576 / w))
  python bench.py --windows-per-step $w --steps $s --warmup 2 --cpu-sample 0 --check none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('windows per step $w policy $e', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done
