"""Reads the kernel trace of tools/probe_fwd_sizes.py: mean duration of the forward kernels per sample count (6 launches each)."""
import csv, sys, glob
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for name in ("mlp_forward_f16_gen", "freq_forward"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
    sizes = (64, 512, 2048, 4096, 8192)
    for i, n in enumerate(sizes):
        v = d[6 * i + 1:6 * i + 6]
        if v:
            print(f"{name:22s} rays {n:5d} x 512: {sum(v) / len(v):8.1f} us")
