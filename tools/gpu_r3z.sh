#!/bin/bash
# general fp16 forward (lnr_f16_fwd_kernel.h): parity + the north-star network leg + duration against the sample count
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_known_answer.py -m gpu -q -k "fp16 or f16 or known or density" > gpurun_out/r3z_tests.log 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/r3z_tests.log
for i in 1 2; do timeout 300 python tools/probe_ns.py 2>&1 | tail -1; done | tee gpurun_out/r3z_probe.log
rm -rf gpurun_out/fwdsizes
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fwdsizes -o p -- python tools/probe_fwd_sizes.py > gpurun_out/fwdsizes.log 2>&1
python tools/fwd_sizes_report.py gpurun_out/fwdsizes | tee gpurun_out/fwdsizes.txt
