#!/bin/bash
# Effective shader clock of conv_mfma_kernel with and without its global loads:
# GRBM_GUI_ACTIVE cycles / kernel duration, per ablation variant (diagnostic).
OUT=gpurun_out/convclk
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 0 1 11 4; do
  export SSD_HIP_LIBRARY=$GRAFT_REPO_ROOT/tf-ssd_amd/csrc/build/ablate/libssd_hip_ab$v.so
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/v$v -o c -- python tests/micro/conv_ablate.py --child > $OUT/v$v.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for v in (0, 1, 11, 4):
    d = "gpurun_out/convclk/v%d" % v
    dur = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    agg = collections.defaultdict(lambda: [0, 0, 0.0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_mfma" not in r["Kernel_Name"]:
                continue
            key = r["Kernel_Name"][:70]
            a = agg[(key, r["Counter_Name"])]
            a[0] += 1
            a[2] += float(r["Counter_Value"])
            a[3] += dur.get(r["Dispatch_Id"], ("", 0))[1]
    print("variant", v)
    for (k, c), a in sorted(agg.items()):
        if c == "GRBM_GUI_ACTIVE":
            print("  %-72s n=%3d  avg %.1f us  cycles %.0f  => %.3f GHz" % (k, a[0], a[3] / a[0] / 1e3, a[2] / a[0], a[2] / a[3] if a[3] else 0))
PY
