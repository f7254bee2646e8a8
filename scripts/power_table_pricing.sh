#!/bin/bash
# Package power while (a) the 9-mer table gather runs alone, (b) the default f16x3 kernel, (c) the same without conv2's MFMAs:
# joules per 4096 windows of conv2 as a table lookup vs on the matrix pipe.
cd "$(dirname "$0")/.."
smp() { for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>/dev/null | grep -i "GPU\[0\]" | grep -i "power\|sclk" | sed 's/^/    /' | tr '\n' ' '; echo; sleep 1.2; done; }
echo "== gather alone"; ./build_variants/probe_gather loop 9 | tail -1 & sleep 3; smp; wait
echo "== f16x3 default"; bash scripts/power_sample.sh f16x3 9 | grep -i "power\|windows/s" | sed 's/^/    /'
echo "== f16x3 without conv2 MFMAs"; GENOMAD_AMD_LIB=build_variants/lib_x3noconv2.so bash scripts/power_sample.sh f16x3 9 | grep -i "power\|windows/s" | sed 's/^/    /'
