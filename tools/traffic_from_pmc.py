"""profiles/traffic.json from the two rocprofv3 PMC passes of tools/pmc.sh (FETCH_SIZE, WRITE_SIZE; per-kernel means)."""
import collections, csv, glob, json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loner_amd.build import sources_digest

def per_kernel(counter):
    # tools/gpu_run.sh pmc:<counter> writes gpurun_out/pmc_<counter>_<tag>/ (tag "product" by default); tools/pmc.sh gpurun_out/pmc_<counter>/
    suffix = sys.argv[2] if len(sys.argv) > 2 else "_product"
    files = glob.glob(f"gpurun_out/pmc_{counter}{suffix}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}

ALIAS = {"encode_backward_kernel": "encode_backward", "encode_forward_kernel<2, false>": "encode_forward", "encode_forward_pair_kernel<2, false>": "encode_forward",
         "encode_forward_pair_kernel<2, true>": "encode_forward_f16", "encode_forward_kernel<2, true>": "encode_forward_f16", "table_grad_reduce2_kernel": "table_grad_reduce",
         "table_grad_reduce_split_kernel": "table_grad_reduce_split",
         "mlp_backward_relu32_kernel": "mlp_backward", "mlp_forward_relu32_kernel": "mlp_forward",
         "mlp_backward_bf3_kernel": "mlp_backward", "mlp_forward_bf3_kernel": "mlp_forward",
         "sum_dx_planes_kernel": "sum_dx_planes", "adam_kernel": "adam", "los_loss_fused_kernel": "los_loss_fused"}
fetch, write = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), bench.py --quick --steps 8 --warmup 4 (tools/gpu_run.sh pmc:FETCH_SIZE pmc:WRITE_SIZE), 1x MI355X",
       "units": "counter values are KiB; bytes = value*1024.  bytes_corrected doubles FETCH_SIZE as MI355X_MICROARCH.md prescribes for gfx950 "
                "(64 B tallied per 128-B request on wide streaming reads; for 8-byte gathers the factor is uncalibrated, so it is an upper bound); "
                "Infinity-Cache hits are counted, not excluded",
       # what the passes measured: bench.py compares this with the sources it runs on and flags a stale file
       "kernel_sources_sha": sources_digest(),
       "commit": (subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("LNR_COMMIT", "unknown")),
       "per_launch": {}}
for k in sorted(set(fetch) | set(write)):
    short = next((v for a, v in ALIAS.items() if a in k), None)
    if short is None:
        continue
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    ent = out["per_launch"].setdefault(short, {"kernels": [], "FETCH_SIZE_KiB": 0.0, "WRITE_SIZE_KiB": 0.0, "bytes_raw": 0.0, "bytes_corrected": 0.0})
    ent["kernels"].append(k[:100])             # an op made of several kernels (encode_backward: one per record format) is their sum
    ent["FETCH_SIZE_KiB"] += f; ent["WRITE_SIZE_KiB"] += w
    ent["bytes_raw"] += (f + w) * 1024.0; ent["bytes_corrected"] += (2.0 * f + w) * 1024.0
    out[short + "_bytes_per_launch"] = ent["bytes_corrected"]
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "profiles/traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k.endswith("_per_launch") and k != "per_launch"}, indent=1))
