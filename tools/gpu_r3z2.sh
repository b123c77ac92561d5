#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/fwdsizes
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fwdsizes -o p -- python tools/probe_fwd_sizes.py > gpurun_out/fwdsizes.log 2>&1
python tools/fwd_sizes_report.py gpurun_out/fwdsizes | tee gpurun_out/fwdsizes.txt
