#!/bin/bash
for i in 1 2; do python tools/superrest_timing.py 100000 12 250 --reserve 2>&1 | grep -E "reserve|resident"; done
python tools/superrest_timing.py 100000 12 250 2>&1 | grep -E "resident"
