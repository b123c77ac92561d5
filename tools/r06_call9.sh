#!/bin/bash
# round 6, call 9: per-kernel times of the 256 x 2 fp16 backward; f16 mode in the driver's short window
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_wide
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wide -o run -- python tools/probe_wide_nets.py --only "freq12 -> 256 x 2" --prec fp16 > gpurun_out/prof_wide.log 2>&1
f=$(find gpurun_out/prof_wide -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | cut -c1-110 | head -16
for st in 20 100; do
  echo "===== f16 mode, --steps $st --warmup 5"
  timeout 300 python bench.py --quick --dtype f16 --steps $st --warmup 5 2>/dev/null | python tools/bench_kernels.py --all | grep -E "ms_per_step|kernel " | head -12
done
