#!/bin/bash
# A/B of two builds on the ABD probe (base = scri_amd/libscri_amd_base.so)
for lib in scri_amd/libscri_amd_base.so scri_amd/libscri_amd.so; do
  echo $lib; SCRI_AMD_LIB_PATH=$PWD/$lib python tools/separable_probe_abd.py ${1:-25000} ${2:-24} 3 2>&1 | grep -v amdgpu.ids | cut -c1-330
done
