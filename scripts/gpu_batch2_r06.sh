#!/bin/bash
# Round 6, batch 2: the GPU suite on the round's kernel, the default bench line (with the main_e2e / metagenome blocks), per-phase cycles.
set -u
out=gpurun_out/r06_batch2; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1
timeout 300 python scripts/tc_check.py 32 t > $out/tc_check.txt 2>&1
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -15 $out/pytest_gpu.txt; tail -22 $out/tc_check.txt; tail -5 $out/bench_default.err; python - <<'PY'
import json
o=json.loads(open('gpurun_out/r06_batch2/bench_default.json').read().strip().splitlines()[-1])
print({k:o.get(k) for k in ('value','ms_per_step','failed','max_abs_dscore','per_rank_device')})
print('roofline', {k:o['roofline'].get(k) for k in ('frac','avg_launch_ms','mfma_passes','traffic_over_algorithmic')})
print('parity', o.get('parity'))
print('main_e2e', json.dumps(o.get('main_e2e'))[:3000])
print('metagenome', json.dumps(o.get('metagenome'))[:1500])
print('cpu', {k:o['cpu_baseline'].get(k) for k in ('value','cores','host_cpus_visible','threads_cap_reason')})
PY
