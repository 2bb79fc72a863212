#!/bin/bash
# fused reduction finalize (training) A/B + uint8 pinned ingest
OUT=gpurun_out/r4w
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_HIP_WARN_STALE_TABLE=0
timeout 1200 python -m pytest tests/test_train.py tests/test_fullsize_gpu.py -m gpu -q -x -s -k "finalized or uint8 or c4_per_gpu or autograd or bucket_events or trainer_entry" 2>&1 | grep -v Warn | tail -8
for f in 0 1 0 1; do
  SSD_TRAIN_FUSED_FINALIZE=$f timeout 900 python bench.py --train --no-cpu-baseline > $OUT/train_f$f.json 2> $OUT/train_f$f.err
  python - <<PY
import json
d=json.loads(open("$OUT/train_f$f.json").read().strip().splitlines()[-1]); print("train fused=$f", round(d["value"]), d["ms_per_step"], d.get("timing_spread"))
PY
done
timeout 600 python tests/micro/train_launch_bound.py 2>&1 | grep -v Warn | tail -9
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print("bench", round(d["value"]), d["ms_per_step"]); print(json.dumps(d["h2d_overlapped_predict"], indent=1))
PY
