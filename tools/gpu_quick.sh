#!/bin/bash
# quick GPU check while iterating on the table-gradient kernels: the kernel parity tests that cover them + one bench line per dtype
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "density or full_size or fp16 or hash_known or frozen or atomic" 2>&1 | tail -5
timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>gpurun_out/b.err | python tools/bench_kernels.py
[ -n "$REPORT" ] && timeout 300 python tools/report_regions.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep "lnr regions" | tail -15
true
