#!/bin/bash
# kernel breakdown + SQ counters of the north-star network's backward (general fp16 kernels)
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/nsbwd gpurun_out/pmc_h1 gpurun_out/pmc_h2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nsbwd -o p -- python tools/probe_gen_bwd.py > gpurun_out/nsbwd.log 2>&1
python - <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob("gpurun_out/nsbwd/*kernel_trace.csv")[0])):
    d[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:9]:
    print(f"{k:62s} n={len(v):3d} mean={sum(v)/len(v):8.1f} us min={min(v):8.1f}")
PY
pass() { local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$name -o p -- python tools/probe_gen_bwd.py > gpurun_out/pmc_$name.log 2>&1
  python tools/pmc_mean.py gpurun_out/pmc_$name mlp_backward_f16_gen
}
pass h1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
pass h2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA
