#!/bin/bash
O=gpurun_out/r05f; mkdir -p $O
export SCRI_AMD_SYNTHESIS_EVAL=1
timeout 900 python -m pytest tests/test_gpu_edge_cases.py -k "evaluation_in_it" -x -q 2>&1 | tail -2
python tools/boost_free_rate.py 2>/dev/null | tail -1 | tee $O/se_default.txt
export SCRI_AMD_LIB_PATH=$PWD/scri_amd/libscri_amd_probes.so
for k in 1 4 5; do SCRI_AMD_SE_KNOCK=$k python tools/boost_free_rate.py 2>/dev/null | tail -1; done | tee $O/se_knock.txt
python tools/probes/synthesis_eval_trace.py 2>&1 | tail -10 | tee $O/se_trace.txt
for sh in "1,1,1" "33,16,34" "40,8,30" "40,4,40" "44,1,40"; do echo "shares $sh: $(SCRI_AMD_SE_SHARES=$sh python tools/boost_free_rate.py 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["kernels_ms"]["gemm_synthesis"], d["ms_per_transform"])')"; done | tee $O/se_shares.txt
echo "ring of four rows (two barriers): $(SCRI_AMD_SE_RING4=1 python tools/boost_free_rate.py 2>/dev/null | tail -1)" | tee -a $O/se_shares.txt
