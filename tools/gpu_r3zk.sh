#!/bin/bash
# timeline of one iteration with ONE keyframe (what a rank of an 8-GPU run sees)
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/kf1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kf1 -o p -- python bench.py --no-cpu-baseline --keyframes 1 --steps 30 --warmup 10 > gpurun_out/kf1.log 2>&1
tail -1 gpurun_out/kf1.log | cut -c1-140
python tools/trace_iteration.py gpurun_out/kf1/p_kernel_trace.csv 25 | tee gpurun_out/kf1_trace.txt
