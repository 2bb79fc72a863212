#!/bin/bash
OUT=gpurun_out/r5h; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { tag=$1; shift
  env "$@" python bench.py --no-cpu-baseline $BARGS > $OUT/$tag.json 2> $OUT/$tag.err
  python - $OUT/$tag.json $tag <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    o = r.get("other_mode") or {}
    print("%-34s %.4f ms/step (min %.4f max %.4f) %.0f img/s | other %.4f" % (sys.argv[2], r["ms_per_step"], r["timing_spread"]["ms_per_step_min"], r["timing_spread"]["ms_per_step_max"], r["value"], o.get("ms_per_step", 0)))
except Exception as e:
    print(sys.argv[2], "failed", e, open(sys.argv[1].replace(".json", ".err")).read()[-500:])
PY
}
BARGS="--train" run train_plain A=1
BARGS="--train" run train_plain_native SSD_BENCH_TRAIN_STREAM=native
BARGS="--train --force-dist" run train_forced_native A=1
BARGS="--train --force-dist" run train_forced_null SSD_BENCH_TRAIN_STREAM=null
BARGS="--train --force-dist" run train_forced_native_b1 SSD_HIP_GRAD_BUCKETS=1
# lanes replaying hipGraphs (single in-order stream each) under three queues
BARGS="--no-h2d" run lanes3_direct A=1
BARGS="--no-h2d" run lanes3_graph SSD_HIP_LANE_GRAPH=1
BARGS="--no-h2d --batch 16" run b16_lanes3_direct A=1
BARGS="--no-h2d --batch 16" run b16_lanes3_graph SSD_HIP_LANE_GRAPH=1
BARGS="--no-h2d --batch 1" run b1_lanes3_direct A=1
BARGS="--no-h2d --batch 1" run b1_lanes3_graph SSD_HIP_LANE_GRAPH=1
