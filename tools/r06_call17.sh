#!/bin/bash
# round 6, call 17: north-star legs, 20 launches per timing
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/probe_ns_sizes.py 2>&1 < /dev/null | grep -E "^[0-9]" > gpurun_out/r06_ns_sizes.txt; cat gpurun_out/r06_ns_sizes.txt
