#!/bin/bash
O=gpurun_out/r05i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_transform_abd.py -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_full_size.py -k "cfg5" -x -q 2>&1 | tail -3
python bench.py --workload cfg5 --steps 5 --warmup 2 --cpu-sample 0 --no-live-pmc > $O/bench_cfg5_shard.json 2> $O/bench_cfg5_shard.err; python - $O/bench_cfg5_shard.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["ms_per_step"], {k:round(v["ms_per_step"],2) for k,v in d["kernels"].items()})
PY
SCRI_AMD_NO_ABD_SIGMA_EVAL=1 python bench.py --workload cfg5 --steps 5 --warmup 2 --cpu-sample 0 --no-live-pmc > $O/bench_cfg5_shard_nosigma.json 2>/dev/null; python - $O/bench_cfg5_shard_nosigma.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["ms_per_step"], {k:round(v["ms_per_step"],2) for k,v in d["kernels"].items()})
PY
