#!/bin/bash
# the training step plain / through the forced single-rank exchange with 1 - 8 gradient buckets, weight gradients on or beside the main stream
OUT=gpurun_out/r5i; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { tag=$1; shift
  env "$@" python bench.py --no-cpu-baseline $BARGS > $OUT/$tag.json 2> $OUT/$tag.err
  python - $OUT/$tag.json $tag <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-34s %.4f ms/step (min %.4f max %.4f) %.0f img/s loss %.5f" % (sys.argv[2], r["ms_per_step"], r["timing_spread"]["ms_per_step_min"], r["timing_spread"]["ms_per_step_max"], r["value"], r["config"]["loss_first_step"]))
except Exception as e:
    print(sys.argv[2], "failed", e, open(sys.argv[1].replace(".json", ".err")).read()[-500:])
PY
}
BARGS="--train" run train_plain A=1
BARGS="--train --force-dist" run forced_b4_side A=1
BARGS="--train --force-dist" run forced_b4_inline SSD_HIP_WGRAD_SIDE_BUCKETS=0
BARGS="--train --force-dist" run forced_b8_side SSD_HIP_GRAD_BUCKETS=8
BARGS="--train --force-dist" run forced_b2_side SSD_HIP_GRAD_BUCKETS=2
BARGS="--train --force-dist" run forced_b1 SSD_HIP_GRAD_BUCKETS=1
BARGS="--train --force-dist --dtype bf16" run forced_b4_side_bf16 A=1
BARGS="--train --dtype bf16" run plain_bf16 A=1
timeout 900 python -m pytest tests/test_train.py -m gpu -q -x 2>&1 | tail -3
