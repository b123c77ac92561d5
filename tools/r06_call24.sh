#!/bin/bash
# round 6, call 24: kernel traces of the one-keyframe loop - non-distributed, sharded with the own RCCL binding (all_reduce / reduce_scatter, front in line)
set +e
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
: > gpurun_out/r06_trace_sharded_world1.txt
for cfg in 0:non-distributed_loop 5:sharded_lnr_comm_all_reduce_front_inline 7:sharded_lnr_comm_reduce_scatter_front_inline; do
  i=${cfg%%:*}; name=${cfg#*:}
  rm -rf gpurun_out/prof_shard_$i
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_shard_$i -o run -- python tools/probe_sharded_overhead.py --steps 40 --warmup 5 --only $i > gpurun_out/prof_shard_$i.log 2>&1 < /dev/null
  grep "ms per iteration" gpurun_out/prof_shard_$i.log
  f=$(find gpurun_out/prof_shard_$i -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/trace_idle.py "$f" --last 30 --title "$name" >> gpurun_out/r06_trace_sharded_world1.txt
done
cat gpurun_out/r06_trace_sharded_world1.txt | grep -E "^---"
timeout 600 python tools/probe_sharded_overhead.py --steps 300 2>&1 < /dev/null | grep "ms per iteration" > gpurun_out/r06_sharded_overhead_world1.txt; cat gpurun_out/r06_sharded_overhead_world1.txt
