#!/bin/bash
# Development: per-dispatch SQ counters (instruction mix, busy/wait cycles) of the table-gradient kernels, two --pmc passes.
export TMPDIR=/tmp
A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES"
B="SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"
bash tools/pmc_sq.sh "$A" mix > /dev/null; python tools/pmc_mean.py gpurun_out/pmc_mix encode_backward_kernel table_grad_reduce encode_forward_kernel
bash tools/pmc_sq.sh "$B" busy > /dev/null; python tools/pmc_mean.py gpurun_out/pmc_busy encode_backward_kernel table_grad_reduce encode_forward_kernel
