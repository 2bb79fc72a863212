"""Summarises rocprofv3 CSV output (kernel stats + PMC FETCH_SIZE / WRITE_SIZE) of one
bench.py run into a small text table that is committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = name.replace("ssd::", "").replace("void ", "")
    return name[:100]


def find(pattern):
    hits = glob.glob(os.path.join(root, "**", pattern), recursive=True)
    return hits[0] if hits else None


stats = find("*kernel_stats.csv")
if stats:
    print("== rocprofv3 --kernel-trace --stats (whole bench.py process: warm-up + autotune + timed + roofline legs)")
    rows = list(csv.DictReader(open(stats)))
    print("%-100s %8s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for r in rows[:40]:
        print("%-100s %8s %12.1f %12.2f %7s" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                 float(r["AverageNs"]) / 1e3, r["Percentage"]))
    fam = defaultdict(lambda: [0, 0.0])
    for r in rows:
        n = short(r["Name"])
        key = n.replace("(anonymous namespace)::", "").split("<")[0].split("(")[0]
        fam[key][0] += int(r["Calls"])
        fam[key][1] += float(r["TotalDurationNs"]) / 1e3
    tot = sum(v[1] for v in fam.values())
    print("\n== by kernel family")
    print("%-40s %8s %12s %12s %7s" % ("family", "calls", "total_us", "avg_us", "pct"))
    for k, (n, us) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print("%-40s %8d %12.1f %12.2f %6.1f%%" % (k, n, us, us / max(n, 1), 100 * us / tot))

# one kernel template may serve several layers (conv tiles): the same table split by launch grid, from the kernel trace
trace = find("*kernel_trace.csv") if stats else None
if trace and os.path.dirname(trace) == os.path.dirname(stats):
    by = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(trace)):
        n = short(r["Kernel_Name"])
        if not n.startswith(("conv_mfma3_kernel", "conv_mfma_kernel", "conv_bf16_kernel", "conv_dma_kernel", "conv_wino_kernel", "(anonymous namespace)::conv_skinny_kernel")):
            continue
        wg = int(r["Workgroup_Size_X"])
        key = (n, "%d x %d workgroups of %d" % (int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]), wg))
        by[key][0] += 1
        by[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("\n== dense-conv tile kernels by launch grid (a template serves several layers; avg_us per grid = per layer)")
    print("%-64s %-34s %6s %10s" % ("kernel", "grid", "calls", "avg_us"))
    for (n, g), (c, us) in sorted(by.items(), key=lambda kv: -kv[1][1])[:24]:
        print("%-64s %-34s %6d %10.2f" % (n.replace("(anonymous namespace)::", "").split("(")[0][:64], g, c, us / c))

traffic = {}      # family -> {counter: KB per launch as reported}
for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = find(os.path.join(tag, "**", "*counter_collection.csv")) or find("*%s*counter_collection.csv" % tag[4])
    cands = glob.glob(os.path.join(root, tag, "**", "*counter_collection.csv"), recursive=True)
    if not cands:
        continue
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(cands[0])):
        if r.get("Counter_Name") != counter:
            continue
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    famagg = defaultdict(lambda: [0, 0.0])
    for k, (n, v) in agg.items():
        key = k.replace("(anonymous namespace)::", "").split("<")[0].split("(")[0]
        famagg[key][0] += n
        famagg[key][1] += v
    for k, (n, v) in famagg.items():
        traffic.setdefault(k, {})[counter] = {"launches": n, "KB_per_launch_reported": v / max(n, 1)}
    print("\n== rocprofv3 --pmc %s by kernel family" % counter)
    for k, (n, v) in sorted(famagg.items(), key=lambda kv: -kv[1][1])[:8]:
        print("%-40s %8d launches %14.0f KB total %12.1f KB/launch" % (k, n, v, v / max(n, 1)))
    print("\n== rocprofv3 --pmc %s (KB as reported; gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x)" % counter)
    print("%-100s %8s %14s %14s" % ("kernel", "calls", "sum_KB", "avg_KB/launch"))
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
        print("%-100s %8d %14.0f %14.1f" % (k, n, v, v / max(n, 1)))

if traffic and os.environ.get("TRAFFIC_JSON"):
    import json
    with open(os.environ["TRAFFIC_JSON"], "w") as fh:
        import hashlib
        import subprocess
        repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        hh = hashlib.sha256()
        for f in sorted(glob.glob(os.path.join(repo, "tf-ssd_amd", "csrc", "*.hip")) +
                        glob.glob(os.path.join(repo, "tf-ssd_amd", "csrc", "*.h"))):
            hh.update(open(f, "rb").read())
        try:
            commit = subprocess.check_output(["git", "-C", repo, "rev-parse", "--short", "HEAD"],
                                             stderr=subprocess.DEVNULL).decode().strip()
        except Exception:
            commit = os.environ.get("SSD_COMMIT")       # no .git on the GPU box: collect.sh passes it
        json.dump({"commit": commit, "kernel_sources_sha16": hh.hexdigest()[:16],
                   "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python bench.py --steps 10 "
                             "--warmup 3 --no-cpu-baseline`, summed per kernel family over the whole process",
                   "note": "FETCH_SIZE on gfx950 reports 1/2 of wide (16 B/lane) coalesced reads: double it "
                           "(MI355X_MICROARCH.md, HBM section); Infinity-Cache hits are counted",
                   "families": traffic}, fh, indent=1)
