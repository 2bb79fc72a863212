#!/bin/bash
OUT=gpurun_out/r4l
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_HIP_WARN_STALE_TABLE=0
for dt in f32 bf16; do
  SSD_HIP_TRAIN_AUTOTUNE=0 python bench.py --train --dtype $dt --no-cpu-baseline > $OUT/train_$dt.json 2>/dev/null
  SSD_HIP_TRAIN_AUTOTUNE=2 python bench.py --train --dtype $dt --no-cpu-baseline > $OUT/train_${dt}_tuned.json 2> $OUT/train_${dt}_tuned.err
done
SSD_HIP_TRAIN_AUTOTUNE=0 python bench.py --train --backbone vgg16 --batch 16 --no-cpu-baseline > $OUT/train_vgg.json 2>/dev/null
SSD_HIP_TRAIN_AUTOTUNE=1 python bench.py --train --backbone vgg16 --batch 16 --no-cpu-baseline > $OUT/train_vgg_tuned.json 2>/dev/null
for f in $OUT/*.json; do python - "$f" <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "%.4f ms/step %.0f img/s"%(r["ms_per_step"], r["value"]), "loss", r["config"]["loss_first_step"], r["config"]["loss_last_step"], "frac %.3f"%r["roofline"]["frac"])
PY
done
grep "train tune" $OUT/train_f32_tuned.err | awk '{m=$(NF-6); b=$(NF-1); sm+=m; sb+=b} END {print "fp32: model picks", sm, "us; best picks", sb, "us over", NR, "shapes"}'
grep "train tune" $OUT/train_bf16_tuned.err | awk '{m=$(NF-6); b=$(NF-1); sm+=m; sb+=b} END {print "bf16: model picks", sm, "us; best picks", sb, "us over", NR, "shapes"}'
grep "wgrad" $OUT/train_f32_tuned.err | head -60
