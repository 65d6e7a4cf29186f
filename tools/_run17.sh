#!/bin/bash
export PYTHONPATH=.
python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()})
print(json.dumps(d['boost_free'])[:900])
"
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
