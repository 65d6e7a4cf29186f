#!/bin/bash
# round 5, second GPU pass: the fused synthesis + evaluation kernel -- parity tests, then the boost-free lines
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_edge_cases.py -k "evaluation_in_it" -x -q > $O/pytest_fused.log 2>&1; echo "fused tests rc $?"; tail -15 $O/pytest_fused.log
timeout 600 python -m pytest tests/test_gpu_guard_regions.py -k "boost-free" -x -q 2>&1 | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for e in 0 1; do
  if [ $e = 1 ]; then export SCRI_AMD_NO_SYNTHESIS_EVAL=1; fi
  python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-live-pmc > $O/bench_cfg3_noeval$e.json 2> $O/bench_cfg3_noeval$e.err
  python - $O/bench_cfg3_noeval$e.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], json.dumps(d.get("boost_free")))
except Exception as e: print("bad line", sys.argv[1], e)
PY
done
unset SCRI_AMD_NO_SYNTHESIS_EVAL
SCRI_AMD_TRACE=1 python tools/superrest_timing.py 100000 12 250 --reserve > $O/superrest_reserve_trace.txt 2>&1; grep -v "us$" $O/superrest_reserve_trace.txt | grep -v "^\[scri_amd\] work space.*hipMalloc [0-9]\.[0-9] ms" | tail -40
