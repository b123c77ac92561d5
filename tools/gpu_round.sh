#!/bin/bash
# Run on the GPU box via gpurun: tests, smoke, bench, optional rocprof. Everything lands in gpurun_out/.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
MODE=${1:-all}
echo "== rocminfo ==" > gpurun_out/env.txt
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6 >> gpurun_out/env.txt
nproc >> gpurun_out/env.txt; lscpu | grep "Model name" >> gpurun_out/env.txt
if [[ "$MODE" == "all" || "$MODE" == "tests" ]]; then
  timeout 1500 python -m pytest tests -m gpu -q -rA --no-header -p no:cacheprovider -s > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
  grep -E "^(PASSED|FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -80
fi
if [[ "$MODE" == "all" || "$MODE" == "smoke" ]]; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
  tail -3 gpurun_out/smoke.log
fi
if [[ "$MODE" == "all" || "$MODE" == "bench" ]]; then
  timeout 400 python bench.py --steps ${STEPS:-30} --warmup ${WARMUP:-10} ${BENCH_ARGS} > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
  tail -4 gpurun_out/bench.log
fi
if [[ "$MODE" == "prof" || "$MODE" == "all+prof" ]]; then
  rm -rf gpurun_out/prof
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r01 -- python bench.py --steps ${STEPS:-30} --warmup ${WARMUP:-10} --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
  echo "prof exit $?" >> gpurun_out/prof_bench.log
  find gpurun_out/prof -name "*kernel_stats*" | head; tail -3 gpurun_out/prof_bench.log
fi
