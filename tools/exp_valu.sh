#!/bin/bash
# Development: VALU instruction counts of encode_backward_kernel per ablation (needs a -DLNR_ABLATE build)
export TMPDIR=/tmp LNR_EXTRA_HIPCC_FLAGS=-DLNR_ABLATE
for d in 0 1 3 7 16 23 31 128; do
  export LNR_X_DBG=$d
  bash tools/pmc_sq.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" valu > /dev/null
  echo "== LNR_X_DBG=$d"; python tools/pmc_mean.py gpurun_out/pmc_valu encode_backward_kernel | grep -E "VALU|SALU|INSTS_LDS"
done
rm -rf gpurun_out/pmc_valu
