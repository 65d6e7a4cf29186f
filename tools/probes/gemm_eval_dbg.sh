#!/bin/bash
# timing experiments on the evaluating product: what its epilogue is made of (results are wrong with any switch set)
for d in ${DBGS:-0 1 2 4 8 10 15}; do
  echo "== SCRI_AMD_GEMM_EVAL_DBG=$d"
  SCRI_AMD_GEMM_EVAL_DBG=$d SCRI_AMD_GEMM_EVAL_STEP=${STEP:-64} python tools/probes/gemm_eval_ab.py 100000 2>&1 | grep -E "^    new"
done
