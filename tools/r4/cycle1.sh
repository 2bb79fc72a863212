#!/bin/bash
# tables at the current build, adopted on the box, then the suite and a per-layer bench (fp32 + bf16)
OUT=gpurun_out/r4v
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/r4/tables.sh > $OUT/tables.log 2>&1
tail -3 $OUT/tables.log
python tools/make_tuning_tables.py --adopt gpurun_out/tables_r4
( time timeout 1500 python -m pytest tests -m gpu -q -rA --durations=10 ) > $OUT/gputest.log 2>&1
echo "pytest rc=$?" >> $OUT/gputest.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $OUT/gputest.log | tail -12
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
timeout 900 python bench.py --lanes 1 --no-overlap --no-other-leg --no-h2d --no-cpu-baseline --layers > $OUT/bench_unc.json 2> $OUT/bench_layers.txt
timeout 900 python bench.py --dtype bf16 --no-cpu-baseline > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
timeout 900 python bench.py --dtype bf16 --lanes 1 --no-overlap --no-other-leg --no-h2d --no-cpu-baseline --layers > $OUT/bench_bf16_unc.json 2> $OUT/bench_bf16_layers.txt
for f in bench bench_unc bench_bf16 bench_bf16_unc; do python - <<PY
import json
d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1]); print("$f", round(d["value"]), d["ms_per_step"], d.get("other_mode",{}).get("ms_per_step"))
PY
done
