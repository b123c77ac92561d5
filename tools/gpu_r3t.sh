#!/bin/bash
export TMPDIR=/tmp
for t in "" u4w8 u3w8 u4w4; do
  if [ -n "$t" ]; then export LNR_LIB_PATH=$PWD/loner_amd/_lib/libloner_hip_$t.so; fi
  echo "== ${t:-u2w8}"
  timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python tools/bench_kernels.py | grep -E "ms_per_step|table_grad"
done
