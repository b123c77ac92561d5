#!/bin/bash
# SQ instruction-mix / busy counters of the encode-backward kernels (scan and binned forms)
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TA|TCP|TD)_[A-Z0-9_]+" | sort -u > gpurun_out/counters_gfx950.txt; wc -l gpurun_out/counters_gfx950.txt
pass() {  # name env counters...
  local name=$1 envs=$2; shift 2
  rm -rf gpurun_out/pmc_$name
  env $envs timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$name -o p -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_$name.log 2>&1
  echo "$name exit $?"
  python tools/pmc_mean.py gpurun_out/pmc_$name encode_backward table_grad_reduce2
}
S1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH"
S2="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
pass scan1 LNR_NO_BINS=1 $S1
pass scan2 LNR_NO_BINS=1 $S2
pass bin1 LNR_BINS_W8=1 $S1
pass bin2 LNR_BINS_W8=1 $S2
