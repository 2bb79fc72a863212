#!/bin/bash
OUT=gpurun_out/r4t
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_bf16_gpu.py -m gpu -q -x -k "stem or fused or bf16 or image_block_split" 2>&1 | tail -4
for f in 0 3 0 3; do
  SSD_STEM_FORM=$f timeout 900 python bench.py --layers --steps 60 > $OUT/fp32_stem$f.json 2> $OUT/fp32_stem$f.txt
  python - <<PY
import json
d=json.loads(open("$OUT/fp32_stem$f.json").read().strip().splitlines()[-1]); print("fp32 stem form=$f", round(d["value"]), d["ms_per_step"])
PY
  grep "stem_fused" $OUT/fp32_stem$f.txt
done
timeout 900 python bench.py --dtype bf16 --layers --steps 60 > $OUT/bf16.json 2> $OUT/bf16.txt
python - <<PY
import json
d=json.loads(open("$OUT/bf16.json").read().strip().splitlines()[-1]); print("bf16", round(d["value"]), d["ms_per_step"])
PY
grep "stem_fused" $OUT/bf16.txt
