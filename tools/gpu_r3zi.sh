#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "render or l1 or inference or model or depth" > gpurun_out/r3zi_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3zi_tests.log
for i in 1 2; do timeout 600 python bench.py --mode render --steps 5 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_scan'], d['kernels_ms_per_scan'])"; done
