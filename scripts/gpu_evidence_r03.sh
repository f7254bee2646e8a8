#!/bin/bash
# Round-3 evidence run on the GPU box (via gpurun): GPU test-suite, un-profiled bench lines, end-to-end runs, one more box of
# the fresh-process hunt.  scripts/gpu_profile_r03.sh holds the rocprofv3 part.   -> gpurun_out/r03e/*
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r03e
mkdir -p $OUT
(timeout 600 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -E "passed|failed|config 2|config 3|seed-43|Error|error" | tail -12) > $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
b() { name=$1; shift; timeout 900 python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "rc=$?" >> $OUT/bench_$name.err; }
b default
b async_steps --async-steps --steps 16 --cpu-sample 0
b power --power --steps 32 --cpu-sample 0 --check none --fast-mode-steps 0
b power_f16c6 --power --precision f16c6 --steps 32 --cpu-sample 0 --check none
b 2048_per_step --windows-per-step 2048 --steps 64 --cpu-sample 0
b 2048_per_step_async --windows-per-step 2048 --steps 64 --cpu-sample 0 --async-steps
b f16c6 --precision f16c6 --steps 16 --cpu-sample 0
b bf16x3 --precision bf16x3 --steps 16 --cpu-sample 0 --check golden
b f16c8 --precision f16c8 --steps 16 --cpu-sample 0 --check golden
b weak --scaling weak --steps 8 --cpu-sample 0
b metagenome_60gbp --workload metagenome --gbp-total 60 --cpu-sample 0
timeout 120 python bench.py --gpus 2 --share-devices --steps 2 --cpu-sample 0 > $OUT/bench_gpus2_share_devices.txt 2>&1; echo "rc=$?" >> $OUT/bench_gpus2_share_devices.txt
(timeout 300 python scripts/real_input_bench.py 600) > $OUT/real_input.txt 2>&1
bash scripts/async_hunt.sh 60 auto > $OUT/async_hunt.txt 2>&1
ls -la $OUT
