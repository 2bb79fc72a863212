#!/bin/bash
OUT=gpurun_out/r4n
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_HIP_WARN_STALE_TABLE=0
( time timeout 1500 python -m pytest tests/test_train.py tests/test_loss.py tests/test_bf16_gpu.py -m gpu -q -x ) > $OUT/t.log 2>&1
grep -E "^E  |passed|failed|real" $OUT/t.log | head
for ws in 1; do for dt in f32; do
  SSD_HIP_TRAIN_WGRAD_STREAM=$ws python bench.py --train --dtype $dt --no-cpu-baseline > $OUT/train_${dt}_ws$ws.json 2>/dev/null
done; done
python bench.py --train --no-cpu-baseline --force-dist > $OUT/train_f32_forcedist.json 2>/dev/null
for f in $OUT/*.json; do python - "$f" <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "%.4f ms/step %.0f img/s"%(r["ms_per_step"], r["value"]), "loss", r["config"]["loss_first_step"], r["config"]["loss_last_step"], "frac %.3f"%r["roofline"]["frac"])
PY
done
