#!/bin/bash
export PYTHONPATH=.
for i in 1 2 3; do
echo "registered: $(python tools/host_mode_rate.py 2>&1 | tail -1 | cut -c20-60)"
echo "pageable:   $(SCRI_AMD_NO_REGISTER=1 python tools/host_mode_rate.py 2>&1 | tail -1 | cut -c20-60)"
done
