#!/bin/bash
# SQ counters of the general fp16 forward (north-star network)
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/pmc_f1 gpurun_out/pmc_f2
pass() { local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$name -o p -- python tools/probe_gen_bwd.py > gpurun_out/pmc_$name.log 2>&1
  python tools/pmc_mean.py gpurun_out/pmc_$name mlp_forward_f16_gen
}
pass f1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
pass f2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA
