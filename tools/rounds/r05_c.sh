#!/bin/bash
# the fused kernel: parity tests, then knock-outs (probe build: results wrong, timing only)
O=gpurun_out/r05c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_edge_cases.py -k "evaluation_in_it" -x -q 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_guard_regions.py -k "boost-free" -x -q 2>&1 | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python tools/boost_free_rate.py 2>/dev/null | tail -1 | tee $O/se_default.txt
SCRI_AMD_NO_SYNTHESIS_EVAL=1 python tools/boost_free_rate.py 2>/dev/null | tail -1 | tee -a $O/se_default.txt
export SCRI_AMD_LIB_PATH=$PWD/scri_amd/libscri_amd_probes.so
for k in 0 1 4 5 8; do SCRI_AMD_SE_KNOCK=$k python tools/boost_free_rate.py 2>/dev/null | tail -1; done | tee $O/se_knock.txt
