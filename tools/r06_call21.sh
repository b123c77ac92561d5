#!/bin/bash
# round 6, call 21: HBM bytes of the fused-frequency north-star legs (PMC FETCH_SIZE / WRITE_SIZE, separate passes, kernel-trace only)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmc_ns_$ctr; rm -rf $out
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o run -- python tools/probe_ns_trace.py 4096 > $out.log 2>&1 < /dev/null
  f=$(find $out -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $ctr <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sys.argv[2]:11s} {k:60s} launches {len(v):3d}  mean {sum(v)/len(v):12.1f} KiB")
PY
done
