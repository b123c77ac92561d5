#!/bin/bash
export TMPDIR=/tmp
for d in ${DBG:-0}; do
  rm -rf gpurun_out/prof_$d
  LNR_DEBUG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$d -o x -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/exp_bench_$d.log 2>&1
done
