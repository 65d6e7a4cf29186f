#!/bin/bash
# Needs the probe build of the library (its switch does not exist in the default one; scri_amd/csrc/env.h):
#   make -C scri_amd/csrc PROBES=1 && export SCRI_AMD_LIB_PATH=$PWD/scri_amd/libscri_amd_probes.so
# rotation kernel: l groups x series length x l range (kernel ms from HIP events, fraction of 8 TB/s)
for L in 16 8; do for N in 100000 1000000; do for G in 1 2 3 4; do
  SCRI_AMD_ROTATE_GROUPS=$G CPU_BASELINE=0 python tools/bench_rotation.py $L $N 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('l<=$L n=$N groups=$G', 'ms', round(d['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3))"
done; done; done
