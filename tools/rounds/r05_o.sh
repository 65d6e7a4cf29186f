#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_edge_cases.py -k "eval_window_statistics or reserved_slab" -x -q 2>&1 | tail -8
