#!/bin/bash
# further seeded sweeps of the random-transformation tests on seeds the suite and the earlier sweeps did not use
O=gpurun_out/r05k; mkdir -p $O
for first in 1000 1500 2000 2500 3000 3500; do
  timeout 900 python tools/fuzz_sweep.py $((first + 500)) $first 2>&1 | grep -E "FAILED|done, failures" | sed "s/^/seeds $first..$((first + 500)): /" >> $O/fuzz_1000_4000.txt
done
for first in 4000 4300; do
  SCRI_AMD_SYNTHESIS_EVAL=1 timeout 900 python tools/fuzz_sweep.py $((first + 300)) $first 2>&1 | grep -E "FAILED|done, failures" | sed "s/^/fused route, seeds $first..$((first + 300)): /" >> $O/fuzz_1000_4000.txt
  SCRI_AMD_FUZZ_AXIS=1 timeout 900 python tools/fuzz_sweep.py $((first + 300)) $first 2>&1 | grep -E "FAILED|done, failures" | sed "s/^/axis boosts, seeds $first..$((first + 300)): /" >> $O/fuzz_1000_4000.txt
  SCRI_AMD_NO_GEMM_EVAL=1 timeout 900 python tools/fuzz_sweep.py $((first + 300)) $first 2>&1 | grep -E "FAILED|done, failures" | sed "s/^/marching route, seeds $first..$((first + 300)): /" >> $O/fuzz_1000_4000.txt
done
cat $O/fuzz_1000_4000.txt
