#!/bin/bash
# A/B of two builds of the library on the bench's secondary lines (boost_free synthesis, rotation): base = scri_amd/libscri_amd_base.so
for lib in scri_amd/libscri_amd_base.so scri_amd/libscri_amd.so; do
  for i in 1 2; do
    SCRI_AMD_LIB_PATH=$PWD/$lib python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-live-pmc 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); b=d['boost_free']; print('$lib', 'cfg3 ms', round(d['ms_per_step'],3), 'boost_free ms', round(b['ms_per_step'],3), 'syn', round(b['synthesis_roofline']['ms_per_launch'],4), round(b['synthesis_roofline']['frac'],3), 'rot', round(d['rotation']['kernel_ms'],4), {k: round(v['ms_per_step'],3) for k,v in d['kernels'].items()})
"
  done
done
