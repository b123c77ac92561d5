"""Per-kernel MEAN (per dispatch) of the counters of a rocprofv3 --pmc run: python tools/pmc_mean.py gpurun_out/pmc_<name> [kernel substring ...]"""
import collections, csv, glob, sys
d = sys.argv[1]
want = sys.argv[2:] or ["encode_backward_kernel", "table_grad_reduce", "encode_forward_kernel"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if not any(w in k for w in want):
        continue
    print(k[:70])
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} mean {sum(v) / len(v):16.0f}   dispatches {len(v)}")
