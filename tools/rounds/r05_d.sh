#!/bin/bash
export SCRI_AMD_LIB_PATH=$PWD/scri_amd/libscri_amd_probes.so
python tools/probes/synthesis_eval_trace.py 2>&1 | tail -12
