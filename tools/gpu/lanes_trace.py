"""Diagnostics: per-kernel durations in the steady three-lane region against the one-at-a-time region (rocprofv3 kernel traces)."""
import csv, glob, re, sys, collections
out = sys.argv[1]


def load(d):
    f = glob.glob(out + "/" + d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def short(n):
    n = n.replace("void ", "").replace("ssd::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", n)[:60]


def steady(rows):
    # the last 20 steps: the region between the 21st-last and the last merge_topk launch
    mt = [i for i, r in enumerate(rows) if "merge_topk" in r["Kernel_Name"]]
    lo, hi = mt[-21] + 1, mt[-1] + 1
    seg = rows[lo:hi]
    return seg, 20


res = {}
for d in ("l1", "l3"):
    seg, n = steady(load(d))
    t0 = min(int(r["Start_Timestamp"]) for r in seg); t1 = max(int(r["End_Timestamp"]) for r in seg)
    ev = []
    for r in seg:
        ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
    ev.sort()
    busy = 0; conc = 0; last = t0; depth = 0; hist = collections.Counter()
    for t, dlt in ev:
        if depth > 0: busy += t - last
        hist[depth] += t - last
        conc += depth * (t - last)
        last = t; depth += dlt
    fam = collections.Counter(); cnt = collections.Counter()
    for r in seg:
        k = short(r["Kernel_Name"]); fam[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[k] += 1
    res[d] = (fam, cnt, n)
    print("%s: wall %.3f ms/step, device busy (union) %.3f ms/step, sum of kernel durations %.3f ms/step, kernels/step %.1f" % (
        d, (t1 - t0) / n / 1e6, busy / n / 1e6, conc / n / 1e6, len(seg) / n))
    print("   time at concurrency depth: " + ", ".join("%d: %.0f%%" % (k, 100.0 * v / (t1 - t0)) for k, v in sorted(hist.items())))
f1, c1, n1 = res["l1"]; f3, c3, n3 = res["l3"]
print("%-62s %9s %9s %7s" % ("kernel (us per step)", "1 lane", "3 lanes", "x"))
for k, v in sorted(f3.items(), key=lambda kv: -kv[1])[:40]:
    a = f1.get(k, 0) / n1 / 1e3; b = v / n3 / 1e3
    print("%-62s %9.1f %9.1f %7.2f" % (k, a, b, b / a if a else 0))
