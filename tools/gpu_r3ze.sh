#!/bin/bash
# round-3 closing run at HEAD: full GPU suite, smoke, default bench (f32) + f16 bench, logs for profiles/
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r3ze_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r3ze_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r3ze_bench_f32.log 2>&1; tail -1 gpurun_out/r3ze_bench_f32.log | cut -c1-160
timeout 900 python bench.py --dtype f16 --no-cpu-baseline > gpurun_out/r3ze_bench_f16.log 2>&1; tail -1 gpurun_out/r3ze_bench_f16.log | cut -c1-160
