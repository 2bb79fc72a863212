#!/bin/bash
OUT=gpurun_out/r5e; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 120 tests/micro/bin/cumask > $OUT/cumask.log 2>&1; cat $OUT/cumask.log
( time timeout 1500 python -m pytest tests/test_convdma_gpu.py -m gpu -q -rA -x -k nets_on ) > $OUT/dmatests.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR|Error|assert|dmab tiles" $OUT/dmatests.log | head -20
run() {  # tag, env..., -- bench args
  tag=$1; shift
  env "$@" python bench.py --no-h2d --no-cpu-baseline --no-other-leg $BARGS > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - $OUT/bench_$tag.json $tag <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %.0f img/s %.4f ms/step  spread %s  cal %s" % (sys.argv[2], r["value"], r["ms_per_step"], [round(v, 4) for v in r["timing_spread"].get("ms_per_step_min_median_max", [r["timing_spread"]["ms_per_step_min"], r["timing_spread"]["ms_per_step_max"]])], (r["config"].get("lane_calibration") or {}).get("two_lanes_used")))
except Exception as e:
    print(sys.argv[2], "failed", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
}
BARGS="--lanes 3" run base3 A=1
BARGS="--lanes 4" run mask4_div GPU_MAX_HW_QUEUES=4 SSD_HIP_LANE_CUMASK=div
BARGS="--lanes 4" run mask4_mod GPU_MAX_HW_QUEUES=4 SSD_HIP_LANE_CUMASK=mod
BARGS="--lanes 4" run nomask4 GPU_MAX_HW_QUEUES=4
BARGS="--lanes 2" run mask2_div GPU_MAX_HW_QUEUES=2 SSD_HIP_LANE_CUMASK=div
BARGS="--lanes 2" run mask2_mod GPU_MAX_HW_QUEUES=2 SSD_HIP_LANE_CUMASK=mod
BARGS="--lanes 3" run mask3_div GPU_MAX_HW_QUEUES=3 SSD_HIP_LANE_CUMASK=div
BARGS="--lanes 4" run mask4_div_q8 GPU_MAX_HW_QUEUES=8 SSD_HIP_LANE_CUMASK=div
