#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/prof_round3.sh r03 2>&1 | tail -12
( time timeout 600 python bench.py ) > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; tail -4 gpurun_out/bench_default.err
timeout 300 python bench.py --dtype f16 --no-cpu-baseline > gpurun_out/bench_f16.log 2> gpurun_out/bench_f16.err
for f in gpurun_out/bench_default.log gpurun_out/bench_f16.log; do python tools/bench_kernels.py --all < $f | head -12; done
