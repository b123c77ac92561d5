#!/bin/bash
# HBM traffic of the dominant kernels from PMC counters (separate passes, kernel-trace only)
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o p -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
  echo "$c exit $?"; find gpurun_out/pmc_$c -name "*.csv" | head -5
done
