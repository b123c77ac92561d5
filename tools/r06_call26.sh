#!/bin/bash
# round 6, call 26: encode_backward with the d/dx term and the next batch's inputs taken over in front of the record copy-out (-DLNR_ENC_DX_BEFORE_COPYOUT=1)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
LIB=$PWD/loner_amd/_lib/libloner_hip_dxe.so
LNR_LIB_PATH=$LIB timeout 900 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider -k "record_partition or bit_reproducible or density_backward or frozen or binned" > gpurun_out/pytest_gpu_subset.log 2>&1 < /dev/null; tail -3 gpurun_out/pytest_gpu_subset.log
for rep in 1 2; do
  for tag in "" dxe; do
    lib=""; [ -n "$tag" ] && lib=$PWD/loner_amd/_lib/libloner_hip_$tag.so
    echo "== ${tag:-product}"
    LNR_LIB_PATH=$lib timeout 300 python bench.py --quick --steps 100 --warmup 10 2>/dev/null | python tools/bench_kernels.py --all | grep -E "ms_per_step|kernel (encode_backward|table_grad_reduce )"
  done
done
