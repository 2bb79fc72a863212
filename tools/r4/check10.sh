#!/bin/bash
OUT=gpurun_out/r4k
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_HIP_WARN_STALE_TABLE=0
timeout 900 python -m pytest tests/test_bf16_gpu.py -m gpu -q -rA -k "forward" > $OUT/t.log 2>&1
grep -E "^E  |passed|failed|bf16 vs fp32|yardstick" $OUT/t.log | cut -c1-330
python bench.py --dtype bf16 --img-size 512 --batch 16 --lanes 1 --no-overlap --no-other-leg --no-cpu-baseline --layers > $OUT/b512_bf16.json 2> $OUT/b512_bf16_layers.txt
python bench.py --dtype bf16 --img-size 512 --batch 16 --no-cpu-baseline > $OUT/b512_bf16_l3.json 2>/dev/null
python bench.py --dtype bf16 --no-cpu-baseline > $OUT/mb_bf16_l3.json 2>/dev/null
for f in $OUT/*.json; do python - "$f" <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "%.4f ms/step %.0f img/s"%(r["ms_per_step"], r["value"]), r["other_mode"] and r["other_mode"]["ms_per_step"], r["gpu_ms_per_step_by_kind"])
PY
done
grep dwproj $OUT/b512_bf16_layers.txt | awk '$4>0.01' | cut -c1-100
