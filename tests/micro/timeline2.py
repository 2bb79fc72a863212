"""Prints a window of a kernel trace (rocprofv3 --kernel-trace csv) with queue ids: two-lane pipelining analysis.
usage: timeline2.py <trace dir> [start_fraction=0.7] [n_kernels=140]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
n = int(sys.argv[3]) if len(sys.argv) > 3 else 140
i0 = int(len(rows) * frac)
# align to a stem kernel
while i0 < len(rows) and "mbv2_stem_kernel" not in rows[i0]["Kernel_Name"]:
    i0 += 1
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + n]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("ssd::", "").replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:40]
    print("%8.1f  +%7.1f  q%-2s s%-3s %s" % (s / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), name))
