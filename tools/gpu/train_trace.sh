#!/bin/bash
# steady-state kernel trace of the training step (cost-model tile choice: no tuning launches in the trace)
OUT=gpurun_out/${TAG:-r5tr}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_HIP_TRAIN_AUTOTUNE=${TUNE:-1}
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python bench.py --train --steps 6 --warmup 3 --repeats 1 > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-200
f=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
python - "$f" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steady region: the last 6 steps = after the last-but-6th adam_kernel
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
print("adam launches", len(adam), "kernels", len(rows))
lo, hi = adam[-7] + 1, adam[-1] + 1
seg = rows[lo:hi]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
busy = 0; cur_s, cur_e = None, None
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("6 steps: wall %.3f ms/step, device busy (union) %.3f ms/step, kernels/step %.0f" % ((t1 - t0) / 6e6, busy / 6e6, len(seg) / 6))
fam = collections.Counter(); cnt = collections.Counter()
for r in seg:
    n = re.sub(r"<.*", "", r["Kernel_Name"].replace("void ssd::", "").replace("(anonymous namespace)::", "")).split("(")[0]
    fam[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[n] += 1
for n, v in fam.most_common(25):
    print("%-40s %7.3f ms/step  %5.0f launches/step  %6.1f us avg" % (n[:40], v / 6e6, cnt[n] / 6, v / cnt[n] / 1e3))
print("sum of kernel durations %.3f ms/step" % (sum(fam.values()) / 6e6))
PY
python tools/gpu/wgrad_list.py $OUT/trace > $OUT/last_step_kernels.txt 2>&1; rm -rf $OUT/trace
