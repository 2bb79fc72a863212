#!/bin/bash
# SSD512 B=16 fp32 under three lanes: which kernel change made the lanes slower (stem form A/B)
OUT=gpurun_out/r4z
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  tag=$1; shift
  timeout 600 env "$@" > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
d=json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); lc=d["config"].get("lane_calibration") or {}
print("%-22s %7.0f img/s %.4f ms | one at a time %.4f ms | lane check: lanes %.3f one %.3f used %s" % ("$tag", d["value"], d["ms_per_step"], (d.get("other_mode") or {}).get("ms_per_step", 0), lc.get("ms_per_step", 0), lc.get("one_lane_ms_per_step", 0), lc.get("two_lanes_used")))
PY
}
A="python bench.py --img-size 512 --batch 16 --no-cpu-baseline --no-h2d"
for rep in 1 2; do
run f32_default$rep $A
run f32_stem0_$rep SSD_STEM_FORM=0 $A
run f32_nosplit$rep SSD_IMAGE_SPLIT=0 $A
run bf16_default$rep $A --dtype bf16
done
