#!/bin/bash
# round 5, first GPU pass: whole GPU suite, the threading test ten times, cfg3 on the three time axes, superrest with / without reserve
O=gpurun_out/r05a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -5 $O/pytest.log
for i in 1 2 3 4 5 6 7 8 9 10; do python -m pytest tests/test_gpu_threading.py -x -q 2>&1 | tail -1; done | tee $O/threading_x10.txt
for ax in uniform jitter sxs; do
  python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-live-pmc --time-axis $ax > $O/bench_cfg3_$ax.json 2> $O/bench_cfg3_$ax.err
  python - $O/bench_cfg3_$ax.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d.get("eval_window"), d.get("boost_free",{}).get("ms_per_step"))
except Exception as e: print("bad line", sys.argv[1], e)
PY
done
python tools/superrest_timing.py 100000 12 250 > $O/superrest_plain.txt 2>&1; tail -3 $O/superrest_plain.txt
python tools/superrest_timing.py 100000 12 250 --reserve > $O/superrest_reserve.txt 2>&1; tail -4 $O/superrest_reserve.txt
