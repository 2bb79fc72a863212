#!/bin/bash
OUT=gpurun_out/r4h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py --dtype bf16 --img-size 512 --batch 16 --lanes 1 --no-overlap --no-other-leg --no-cpu-baseline --layers > $OUT/b512_bf16.json 2> $OUT/b512_bf16_layers.txt
python bench.py --img-size 512 --batch 16 --lanes 1 --no-overlap --no-other-leg --no-cpu-baseline --layers > $OUT/b512_f32.json 2> $OUT/b512_f32_layers.txt
python bench.py --dtype bf16 --backbone vgg16 --lanes 1 --no-overlap --no-other-leg --no-cpu-baseline --layers > $OUT/vgg_bf16.json 2> $OUT/vgg_bf16_layers.txt
python bench.py --dtype bf16 --lanes 1 --no-overlap --no-other-leg --no-cpu-baseline --layers > $OUT/mb_bf16.json 2> $OUT/mb_bf16_layers.txt
for f in $OUT/*.json; do python - "$f" <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "%.4f ms/step %.0f img/s"%(r["ms_per_step"], r["value"]), r["gpu_ms_per_step_by_kind"])
PY
done
