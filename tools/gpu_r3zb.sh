#!/bin/bash
# full GPU suite + default bench + north-star probe after the general fp16 kernel rewrite
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r3zb_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3zb_tests.log
timeout 600 python bench.py > gpurun_out/r3zb_bench.log 2>&1; tail -1 gpurun_out/r3zb_bench.log | cut -c1-600
for i in 1 2; do timeout 300 python tools/probe_ns.py 2>&1 | tail -1 | cut -c100-330; done
