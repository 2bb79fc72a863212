#!/bin/bash
OUT=gpurun_out/r4i
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -k "pinned or fused_softmax or two_lanes" > $OUT/t.log 2>&1
grep -n "^E  \|Error\|passed\|failed" $OUT/t.log | head
python bench.py --no-cpu-baseline > $OUT/bench_h2d.json 2> $OUT/bench_h2d.err
for cap in 1 2 4; do SSD_IMAGE_GROUPS_CAP=$cap python bench.py --no-cpu-baseline --no-other-leg > $OUT/bench_cap$cap.json 2>/dev/null; done
for cap in 1 2; do SSD_IMAGE_GROUPS_CAP=$cap python bench.py --no-cpu-baseline --no-other-leg --lanes 4 > $OUT/bench_l4_cap$cap.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4i/bench_*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    print(f, "%.4f ms/step %.0f img/s"%(r["ms_per_step"], r["value"]), r.get("h2d_overlapped_predict"))
PY
tail -n 3 $OUT/bench_h2d.err
