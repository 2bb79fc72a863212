"""Per-kernel SQ counter summary (rocprofv3 --pmc ... counter_collection.csv)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
files = glob.glob(os.path.join(root, "sq", "**", "*counter_collection.csv"), recursive=True)
if not files:
    print("no counter_collection.csv under", root)
    sys.exit(0)
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for r in csv.DictReader(open(files[0])):
    k = r["Kernel_Name"].replace("ssd::", "").replace("void ", "").replace("(anonymous namespace)::", "")
    k = k.split("(")[0][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
cols = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
        "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU_MFMA_MOPS_F32"]
print("%-70s %6s %s" % ("kernel", "calls", " ".join("%14s" % c.replace("SQ_", "")[:14] for c in cols)))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:40]:
    print("%-70s %6d %s" % (k, len(calls[k]), " ".join("%14.3g" % v.get(c, 0) for c in cols)))
print("\nderived: mfma_busy/busy_cycles (share of time the MFMA pipe is busy, per SE-aggregated SQ), wait_any/wave_cycles")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:40]:
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    bc = v.get("SQ_BUSY_CYCLES", 0) or 1
    print("%-70s mfma_busy/busy %.3f  wait_any/wave %.3f  wait_inst/wave %.3f  active/wave %.3f  lds_conflict/wave %.4f" % (
        k, v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / bc, v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc,
        v.get("SQ_ACTIVE_INST_ANY", 0) / wc, v.get("SQ_LDS_BANK_CONFLICT", 0) / wc))
