#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mapping.py -m gpu -q -x -k "backward or density or partition or mapping or pose" > gpurun_out/r3zl_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3zl_tests.log
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {k:d['kernels_ms'].get(k) for k in ('encode_backward','table_grad_reduce','encode_forward')})"; done
