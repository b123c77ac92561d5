#!/bin/bash
# A/B of an encode_forward change: forward parity tests + default bench (kernels_ms) + render leg
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_known_answer.py -m gpu -q -x -k "forward or known or encode or density or render" > gpurun_out/r3zc_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3zc_tests.log
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {k:d['kernels_ms'][k] for k in ('encode_forward','encode_backward','table_grad_reduce')}, d['render']['ms_per_scan'] if 'render' in d else None)"; done
