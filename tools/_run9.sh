mkdir -p gpurun_out/r02i
python tools/pipeline_check.py 2>&1 | tail -1
python tools/host_mode_rate.py 2>&1 | tail -1 > gpurun_out/r02i/host_rate_registered.txt
SCRI_AMD_NO_REGISTER=1 python tools/host_mode_rate.py 2>&1 | tail -1 > gpurun_out/r02i/host_rate_pageable.txt
for P in 3 4 8 12 16; do SCRI_AMD_PIPELINE_PIECES=$P python tools/host_mode_rate.py 2>&1 | tail -1 > gpurun_out/r02i/host_rate_registered_p$P.txt; done
for f in gpurun_out/r02i/host_rate_*.txt; do echo $f; cut -c1-90 $f; done
SCRI_AMD_TRACE=1 python tools/host_mode_rate.py 2>&1 | grep "scri_amd" | tail -60 > gpurun_out/r02i/trace.txt
