#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_charges.py tests/test_gpu_superrest.py tests/test_gpu_device_resident.py tests/test_golden.py -x -q 2>&1 | tail -3
