#!/bin/bash
# fresh finalize-time tuning (shipped tables ignored): image kernel race with / without the split form; bf16 WL forms
OUT=gpurun_out/r4r
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_HIP_IGNORE_SHIPPED=1
for sp in 0 1 0 1; do
  SSD_IMAGE_SPLIT=$sp timeout 900 python bench.py --layers --steps 60 > $OUT/fp32_split$sp.json 2> $OUT/fp32_split$sp.txt
  python - <<PY
import json
d=json.loads(open("$OUT/fp32_split$sp.json").read().strip().splitlines()[-1]); print("fp32 split=$sp", round(d["value"]), d["ms_per_step"])
PY
  grep "_fused" $OUT/fp32_split$sp.txt | awk '{print $1,$3,$4}' | tr '\n' ';'; echo
done
for wl in 1 3 1 3; do
  SSD_IMAGE_WL=$wl timeout 900 python bench.py --dtype bf16 --layers --steps 60 > $OUT/bf16_wl$wl.json 2> $OUT/bf16_wl$wl.txt
  python - <<PY
import json
d=json.loads(open("$OUT/bf16_wl$wl.json").read().strip().splitlines()[-1]); print("bf16 WL=$wl", round(d["value"]), d["ms_per_step"])
PY
  grep "_fused\|1[456]_expand\|1[456]_dwproj" $OUT/bf16_wl$wl.txt | awk '{print $1,$3,$4}' | tr '\n' ';'; echo
done
