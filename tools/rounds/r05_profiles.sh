#!/bin/bash
# the committed profile set of the final build: rocprofv3 --kernel-trace --stats + the two PMC passes of the default bench command, the default line
bash tools/run_profiles.sh > gpurun_out/run_profiles.log 2>&1
tail -1 gpurun_out/bench_default.log | cut -c1-300
python3 bench.py --workload cfg5 --steps 5 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 > gpurun_out/bench_cfg5_shard_line.json
