#!/bin/bash
# rocprofv3 evidence for round 3 (run on the GPU box via gpurun): kernel stats of the bench command, PMC traffic passes, SQ counters,
# the same for the inference leg.  Counters are collected in their own runs with --kernel-trace only.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
TAG=${1:-r03}
rm -rf gpurun_out/prof_$TAG gpurun_out/prof_${TAG}_render
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/prof_${TAG}_bench.log 2>&1
echo "stats exit $?"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_render -o ${TAG}_render -- python bench.py --mode render --steps 2 --warmup 10 > gpurun_out/prof_${TAG}_render_bench.log 2>&1
echo "render stats exit $?"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o p -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
  echo "$c exit $?"
done
rm -rf gpurun_out/pmc_sq
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_sq -o p -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_sq.log 2>&1
echo "sq exit $?"
find gpurun_out/prof_$TAG gpurun_out/prof_${TAG}_render -name "*kernel_stats*"; find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_sq -name "*counter_collection.csv" | head
