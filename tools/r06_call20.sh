#!/bin/bash
# round 6, call 20: encode_backward built for six waves per SIMD (80 registers, spills) with a 40 KB staging buffer for the x-pair launch
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {   # run <name> <lib tag> <env...>
  name=$1; tag=$2; shift 2
  rm -rf gpurun_out/prof_$name
  lib=""; [ -n "$tag" ] && lib=$PWD/loner_amd/_lib/libloner_hip_$tag.so
  env "$@" LNR_LIB_PATH=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -o run -- python bench.py --quick --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_$name.log 2>&1 < /dev/null
  echo "== $name"; grep -E "encode_backward|table_grad_reduce2" gpurun_out/prof_$name/run_kernel_stats.csv | awk -F'",' '{print substr($1,1,48), $2, $4}' | tr -d '"'
}
run base "" LNR_DUMMY=1
run w6 w6 LNR_ENC_BWD_XP_STAGE=2560
run w6_nostage w6 LNR_DUMMY=1
