#!/bin/bash
# refresh the kernel-stats summaries of the bench command and of the inference leg at HEAD (profiles/r03_kernel_stats.csv, r03_render_kernel_stats.csv)
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/prof_r03h gpurun_out/prof_r03h_render
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03h -o r03 -- python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/prof_r03h_bench.log 2>&1; echo "stats exit $?"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03h_render -o r03_render -- python bench.py --mode render --steps 2 --warmup 10 > gpurun_out/prof_r03h_render_bench.log 2>&1; echo "render stats exit $?"
find gpurun_out/prof_r03h gpurun_out/prof_r03h_render -name "*kernel_stats*"
tail -1 gpurun_out/prof_r03h_bench.log | cut -c1-200
