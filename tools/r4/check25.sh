#!/bin/bash
# buffer loads in the fp32-MFMA tiles: parity + training / inference A/B against the committed numbers
OUT=gpurun_out/r4b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_HIP_WARN_STALE_TABLE=0
timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_train.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do
timeout 600 python bench.py --train --no-cpu-baseline > $OUT/train$i.json 2>/dev/null
timeout 600 python bench.py --train --backbone vgg16 --batch 16 --no-cpu-baseline > $OUT/trainv$i.json 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --no-h2d > $OUT/inf$i.json 2>/dev/null
python - <<PY
import json
for t in ("train$i", "trainv$i", "inf$i"):
    d=json.loads(open("$OUT/%s.json" % t).read().strip().splitlines()[-1]); print(t, round(d["value"]), d["ms_per_step"])
PY
done
