#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/prof_t
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_t -o t -- python bench.py --steps 30 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
python tools/trace_iteration.py $(find gpurun_out/prof_t -name "*kernel_trace.csv") 20
