#!/bin/bash
export PYTHONPATH=.
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 1700 python -m pytest tests -m gpu -x -q -k "device_resident or edge or kernels or golden" 2>&1 | tail -3
