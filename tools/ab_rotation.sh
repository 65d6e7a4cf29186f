#!/bin/bash
# A/B of library builds on the rotation line (alternating, 3 rounds): tools/ab_rotation.sh lib1.so lib2.so ...
for r in 1 2 3; do for lib in "$@"; do for L in 16 8; do
  SCRI_AMD_LIB_PATH=$PWD/$lib CPU_BASELINE=0 python tools/bench_rotation.py $L 100000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib l<=$L', 'ms', round(d['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3))"
done; done; done
