"""Prints the kernel timeline of the LAST step of a traced bench.py run (rocprofv3 --kernel-trace csv)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at mbv2_stem_kernel
starts = [i for i, r in enumerate(rows) if "mbv2_stem_kernel" in r["Kernel_Name"]]
spans = [(int(rows[starts[i + 1]]["Start_Timestamp"]) - int(rows[starts[i]]["Start_Timestamp"])) / 1e3 for i in range(len(starts) - 1)]
print("step spans (us):", " ".join("%.0f" % v for v in spans))
if len(sys.argv) > 2:
    k = int(sys.argv[2])
else:       # the shortest step = a graph-replayed one (the per-layer timing leg has event gaps)
    k = min(range(len(spans)), key=lambda i: spans[i])
i0 = starts[k]
i1 = starts[k + 1]
t0 = int(rows[i0]["Start_Timestamp"])
last_end = t0
busy = 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("ssd::", "").replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:46]
    print("%8.1f us  +%7.1f us  q%-3s %s" % (s / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name))
print("step span %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
