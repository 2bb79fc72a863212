#!/bin/bash
# run-to-run spread of the headline line (fresh process each): N x bench.py, value + lane check
N=${1:-12}; shift
OUT=gpurun_out/${TAG:-rep}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in $(seq 1 $N); do
  timeout 600 python bench.py --no-cpu-baseline --no-h2d "$@" > $OUT/b$i.json 2> $OUT/b$i.err
  python - <<PY
import json
d=json.loads(open("$OUT/b$i.json").read().strip().splitlines()[-1]); lc=d["config"].get("lane_calibration") or {}
print("run %2d  %7.0f img/s %.4f ms | one at a time %.4f | lanes used %s attempts %s" % ($i, d["value"], d["ms_per_step"], (d.get("other_mode") or {}).get("ms_per_step", 0), lc.get("two_lanes_used"), lc.get("attempts_ms_per_step")))
PY
done
