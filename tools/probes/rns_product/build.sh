#!/bin/bash
# Builds the two stand-alone programs of this experiment next to this script (the library itself is untouched):
#   ceiling : register-only loop -- what the int8 matrix pipe + the fp64 fold could reach per output
#   check   : residue planes + product kernel against the library's fp64 3M product and a long-double host product
set -e
cd "$(dirname "$0")/../../.."
make -C scri_amd/csrc kernels_gemm.o >/dev/null
D=tools/probes/rns_product
F="--offload-arch=gfx950 -O3 -std=c++17 -Iscri_amd/csrc -Iinclude -Wno-unused-result -Wno-unused-value"
hipcc $F $D/ceiling.hip -o $D/ceiling
hipcc $F -fPIC -c $D/kernels_gemm_rns.hip -o $D/kernels_gemm_rns.o
hipcc $F -c $D/check.hip -o $D/check.o
hipcc --offload-arch=gfx950 $D/check.o $D/kernels_gemm_rns.o scri_amd/csrc/kernels_gemm.o -o $D/check
