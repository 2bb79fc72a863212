#!/bin/bash
# where does the forced single-rank exchange lose its millisecond?
OUT=gpurun_out/r5g; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { tag=$1; shift
  env "$@" python bench.py --train --no-cpu-baseline $BARGS > $OUT/train_$tag.json 2> $OUT/train_$tag.err
  python - $OUT/train_$tag.json $tag <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-34s %.3f ms/step (min %.3f max %.3f) loss %.4f" % (sys.argv[2], r["ms_per_step"], r["timing_spread"]["ms_per_step_min"], r["timing_spread"]["ms_per_step_max"], r["config"]["loss_first_step"]))
except Exception as e:
    print(sys.argv[2], "failed", e, open(sys.argv[1].replace(".json", ".err")).read()[-500:])
PY
}
BARGS="" run plain A=1
BARGS="" run plain_again A=1
BARGS="--force-dist" run forced_b4 A=1
BARGS="--force-dist" run forced_b1 SSD_HIP_GRAD_BUCKETS=1
BARGS="--force-dist" run forced_b2 SSD_HIP_GRAD_BUCKETS=2
BARGS="--force-dist" run forced_b8 SSD_HIP_GRAD_BUCKETS=8
BARGS="--force-dist" run forced_b4_nocommstream SSD_HIP_COMM_STREAM=0
BARGS="--force-dist" run forced_b4_q8 GPU_MAX_HW_QUEUES=8
BARGS="--force-dist" run forced_b4_q2 GPU_MAX_HW_QUEUES=2
BARGS="" run plain_q8 GPU_MAX_HW_QUEUES=8
