#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -s -k "fp16_mode_general" 2>&1 | grep -E "^fp16|passed|failed|assert " | tail -14
