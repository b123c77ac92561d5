#!/bin/bash
# round 6, call 12: fused frequency forward with the next step's features evaluated behind the current step's MFMAs
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider -k "fp16 or freq or f16" > gpurun_out/pytest_gpu_subset.log 2>&1; tail -4 gpurun_out/pytest_gpu_subset.log
for i in 1 2 3; do timeout 600 python tools/probe_ns.py 2>&1 | tail -1; done > gpurun_out/r06_ns_side.txt; cat gpurun_out/r06_ns_side.txt
