#!/bin/bash
# round 6, call 28: encode_backward's batch loop specialised on the point-source kind (requests are loads only)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider -k "record_partition or bit_reproducible or density or frozen or binned or ragged or rays_form" > gpurun_out/pytest_gpu_subset.log 2>&1 < /dev/null; tail -3 gpurun_out/pytest_gpu_subset.log
for rep in 1 2; do
  timeout 300 python bench.py --quick --steps 100 --warmup 10 2>/dev/null | python tools/bench_kernels.py --all | grep -E "ms_per_step|kernel (encode_backward|table_grad_reduce )"
done
