#!/bin/bash
export TMPDIR=/tmp
A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES"
B="SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"
bash tools/pmc_sq.sh "$A" a_compact > /dev/null; python tools/pmc_mean.py gpurun_out/pmc_a_compact encode_backward_kernel
bash tools/pmc_sq.sh "$B" b_compact > /dev/null; python tools/pmc_mean.py gpurun_out/pmc_b_compact encode_backward_kernel
export LNR_X_NOCOMPACT=1
bash tools/pmc_sq.sh "$A" a_ident > /dev/null; python tools/pmc_mean.py gpurun_out/pmc_a_ident encode_backward_kernel
bash tools/pmc_sq.sh "$B" b_ident > /dev/null; python tools/pmc_mean.py gpurun_out/pmc_b_ident encode_backward_kernel
rm -rf gpurun_out/pmc_a_* gpurun_out/pmc_b_*
