#!/bin/bash
# round 6, call 18: the fused forward at one wave per SIMD (how much of a wave's time the second wave covers)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in "" 1; do
  echo "LNR_F16_FWD_ONE_PER_CU=$v"
  if [ -n "$v" ]; then export LNR_F16_FWD_ONE_PER_CU=1; fi
  timeout 300 python tools/probe_ns_trace.py 4096 2>&1 < /dev/null | grep -E "^[0-9]"
  timeout 300 python tools/probe_ns_trace.py 16384 2>&1 < /dev/null | grep -E "^[0-9]"
done
