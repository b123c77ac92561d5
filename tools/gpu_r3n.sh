#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  LNR_BINS=1 LNR_BINS_W8=1 timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o p -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
  echo "$c exit $?"
done
python tools/traffic_from_pmc.py gpurun_out/traffic_binned.json | grep -E "encode_backward|table_grad"
python tools/pmc_mean.py gpurun_out/pmc_WRITE_SIZE encode_backward
python tools/pmc_mean.py gpurun_out/pmc_FETCH_SIZE encode_backward
