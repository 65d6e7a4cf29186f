#!/bin/bash
export PYTHONPATH=.
python tools/host_mode_rate.py 2>&1 | tail -1 | cut -c1-70
for p in 12 16 24; do echo "pieces $p: $(SCRI_AMD_PIPELINE_PIECES=$p python tools/host_mode_rate.py 2>&1 | tail -1 | cut -c1-60)"; done
python tools/host_mode_rate.py > gpurun_out/host_mode_rate.txt 2>&1; SCRI_AMD_NO_REGISTER=1 python tools/host_mode_rate.py > gpurun_out/host_mode_rate_pageable.txt 2>&1
