#!/bin/bash
# whole-image blocks under three lanes: ticketed in-launch combine vs the combine launch (fp32 image form: SSD_IMAGE_SPLIT=0)
OUT=gpurun_out/r4x
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_HIP_WARN_STALE_TABLE=0
run() {  # tag, env..., args
  tag=$1; shift
  timeout 600 env "$@" > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
d=json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); print("%-28s %7.0f img/s %.4f ms | one at a time %.4f ms" % ("$tag", d["value"], d["ms_per_step"], (d.get("other_mode") or {}).get("ms_per_step", 0)))
PY
}
for rep in 1 2; do
run split_base$rep python bench.py --no-cpu-baseline --no-h2d
run split_ticket$rep python bench.py --no-cpu-baseline --no-h2d --opt image_ticket=1
run fp32_base$rep SSD_IMAGE_SPLIT=0 python bench.py --no-cpu-baseline --no-h2d
run fp32_ticket$rep SSD_IMAGE_SPLIT=0 python bench.py --no-cpu-baseline --no-h2d --opt image_ticket=1
done
