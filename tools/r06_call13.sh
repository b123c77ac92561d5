#!/bin/bash
# round 6, call 13: which kernels the north-star legs are made of, at 64 K and 2 M samples
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
for rays in 128 4096; do
  rm -rf gpurun_out/prof_ns_$rays
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ns_$rays -o run -- python tools/probe_ns_trace.py $rays > gpurun_out/prof_ns_$rays.log 2>&1 < /dev/null
  grep -E "^[0-9]+ " gpurun_out/prof_ns_$rays.log
  f=$(find gpurun_out/prof_ns_$rays -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -14 "$f" | awk -F'",' '{print substr($1,1,60), $2, $3, $4}' | tr -d '"'
done
