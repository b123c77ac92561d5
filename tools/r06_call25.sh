#!/bin/bash
# round 6, call 25: shared phase part / no clearing / scalar base pointers in the fused forward
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider -k "fp16 or freq or f16 or general or fused_frequency" > gpurun_out/pytest_gpu_subset.log 2>&1 < /dev/null; tail -3 gpurun_out/pytest_gpu_subset.log
timeout 300 python tools/probe_ns_sizes.py 2>&1 < /dev/null | grep -E "^[0-9]" > gpurun_out/r06_ns_sizes.txt; cat gpurun_out/r06_ns_sizes.txt
