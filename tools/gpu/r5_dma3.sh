#!/bin/bash
OUT=gpurun_out/r5d; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python -m pytest tests/test_convdma_gpu.py -m gpu -q -rA -x ) > $OUT/dmatests.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR|Error|assert|dmab tiles" $OUT/dmatests.log | head -30
export SSD_HIP_IGNORE_SHIPPED=1 SSD_HIP_WARN_STALE_TABLE=0
for cfg in "--dtype bf16" "--dtype bf16 --backbone vgg16"; do
  tag=$(echo "$cfg" | tr -d ' -' )
  python bench.py $cfg --no-h2d --no-cpu-baseline > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - $OUT/bench_$tag.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "%.0f img/s %.4f ms/step | one at a time %.4f ms" % (r["value"], r["ms_per_step"], r["other_mode"]["ms_per_step"]))
PY
done
