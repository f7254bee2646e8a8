#!/bin/bash
# Power / clock samples of GPU 0 while the fused kernel of a precision runs: scripts/power_sample.sh [precision] [seconds]
PREC=${1:-f16c6}; SECS=${2:-12}
cd "$(dirname "$0")/.."
rocm-smi --showmaxpower --showpowerprofile 2>/dev/null | grep -v "^=\|^$" | head -8
python - "$PREC" "$SECS" <<'PY' &
import sys, time
sys.path.insert(0, '.')
from genomad_amd import synthetic
from genomad_amd.engine import NNEngine
prec, secs = sys.argv[1], float(sys.argv[2])
eng = NNEngine(0, synthetic.synth_weights(), chunk=4096)
n = 16384
bases, scores = eng.alloc(n * 6000), eng.alloc(n * 12)
eng.synth_windows_dev(0, n, bases.ptr); eng.sync()
t0 = time.time(); it = 0
while time.time() - t0 < secs:
    eng.classify_dev(bases.ptr, n, scores.ptr, prec); eng.sync(); it += 1
print(f"{prec}: {it * n / (time.time() - t0):.0f} windows/s over {secs:.0f} s", flush=True)
PY
sleep 4
for i in 1 2 3 4; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "GPU\[0\]" | grep -i "power\|sclk\|junction\|mclk" | sed 's/^/  /'; echo "  --"; sleep 1.5; done
wait
