"""Kernel sequence of the LAST training step of a rocprofv3 kernel trace (name, grid, workgroup, duration): which layer's launch costs what.
usage: python tools/gpu/wgrad_list.py <trace dir>"""
import csv, sys, re, glob
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
seg = rows[adam[-2] + 1: adam[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    n = re.sub(r"void ssd::|\(anonymous namespace\)::|ssd::", "", r["Kernel_Name"])
    n = re.sub(r"\(.*", "", n)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("%9.1f us  %-46s grid %6s %5s wg %4s  %7.1f us" % ((int(r["Start_Timestamp"]) - t0) / 1e3, n[:46], r.get("Grid_Size_X", r.get("Grid_Size")),
                                                               r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size")), d))
