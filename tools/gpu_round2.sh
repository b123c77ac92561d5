#!/bin/bash
# Run on the GPU box via gpurun: tests, smoke, bench (both dtypes).  Everything lands in gpurun_out/.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
MODE=${1:-all}
if [[ "$MODE" == *tests* || "$MODE" == "all" ]]; then
  timeout ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -q -rA --no-header -p no:cacheprovider -s ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -40
fi
if [[ "$MODE" == *smoke* || "$MODE" == "all" ]]; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
  tail -3 gpurun_out/smoke.log
fi
if [[ "$MODE" == *bench* || "$MODE" == "all" ]]; then
  timeout 400 python bench.py --steps ${STEPS:-30} --warmup ${WARMUP:-10} --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/bench_f32.log 2>gpurun_out/bench_f32.err; echo "bench exit $?" >> gpurun_out/bench_f32.log
  tail -2 gpurun_out/bench_f32.log | cut -c1-1500
  timeout 400 python bench.py --steps ${STEPS:-30} --warmup ${WARMUP:-10} --no-cpu-baseline --dtype f16 ${BENCH_ARGS} > gpurun_out/bench_f16.log 2>gpurun_out/bench_f16.err; echo "bench exit $?" >> gpurun_out/bench_f16.log
  tail -2 gpurun_out/bench_f16.log | cut -c1-1500
fi
