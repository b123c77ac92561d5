#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_known_answer.py -m gpu -q -x -p no:cacheprovider -s -k "fp16 or density_forward or density_backward" 2>&1 | grep -E "^fp16|passed|failed|Error" | tail -12
python tools/probe_ns.py 2>/dev/null | tail -1
