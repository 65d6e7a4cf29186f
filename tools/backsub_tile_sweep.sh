#!/bin/bash
# Needs the probe build of the library (its switch does not exist in the default one; scri_amd/csrc/env.h):
#   make -C scri_amd/csrc PROBES=1 && export SCRI_AMD_LIB_PATH=$PWD/scri_amd/libscri_amd_probes.so
# back substitution of the cfg3 bench: tile length vs kernel time (HIP events), SCRI_AMD_SPLINE_TILE_BWD
for t in 0 172 200 257 290 320 400 513; do
  for i in 1 2; do
    SCRI_AMD_SPLINE_TILE_BWD=$t python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-live-pmc 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('tile $t', 'cfg3 ms', round(d['ms_per_step'],3), 'backward', round(d['kernels']['spline_backward']['ms_per_step'],4), 'boost_free backward', round(d['boost_free']['kernels']['spline_backward'],4))
"
  done
done
