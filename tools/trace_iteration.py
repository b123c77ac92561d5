"""Timeline of one steady-state mapping iteration from a rocprofv3 --kernel-trace CSV (start offset, duration in us, queue, kernel):
    python tools/trace_iteration.py gpurun_out/prof_x/x_kernel_trace.csv [iteration index]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(("void encode_forward_kernel<2, false>", "void encode_forward_pair_kernel<2, false>")) and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < 600000]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
seg = rows[starts[k]:starts[k + 1]]
t0 = int(seg[0]["Start_Timestamp"])
queues = {}
for r in seg:
    q = queues.setdefault(r.get("Queue_Id", "?"), len(queues) + 1)
    print("%8.1f %8.1f  q%d  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, q, r["Kernel_Name"][:70]))
print("iteration span %.1f us" % ((int(rows[starts[k + 1]]["Start_Timestamp"]) - t0) / 1e3))
