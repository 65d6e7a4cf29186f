#!/bin/bash
# Needs the probe build of the library (the switches below do not exist in the default one; scri_amd/csrc/env.h):
#   make -C scri_amd/csrc PROBES=1 && export SCRI_AMD_LIB_PATH=$PWD/scri_amd/libscri_amd_probes.so
make -C scri_amd/csrc PROBES=1 -j8 >/dev/null && export SCRI_AMD_LIB_PATH=$PWD/scri_amd/libscri_amd_probes.so
# timing experiments on the evaluating product: what its epilogue is made of (results are wrong with any switch set)
for d in ${DBGS:-0 1 2 4 8 10 15}; do
  echo "== SCRI_AMD_GEMM_EVAL_DBG=$d"
  SCRI_AMD_GEMM_EVAL_DBG=$d SCRI_AMD_GEMM_EVAL_STEP=${STEP:-64} python tools/probes/gemm_eval_ab.py 100000 2>&1 | grep -E "^    new"
done
