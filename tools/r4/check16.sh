#!/bin/bash
# image16 kernel with LDS-staged weights: split form (fp32 mode) and bf16 form, correctness + timing
OUT=gpurun_out/r4q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== split form correctness"
SSD_IMAGE_SPLIT=1 timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -k "image_block_kernel" 2>&1 | tail -5
echo "== bf16 tests, WL=3"
SSD_IMAGE_WL=3 timeout 900 python -m pytest tests/test_bf16_gpu.py -m gpu -q 2>&1 | tail -5
echo "== fp32 A/B split=0"
SSD_IMAGE_SPLIT=0 timeout 600 python tests/micro/imgblock_ab.py 64 2>&1 | tail -45 > $OUT/ab_split0.txt
echo "== fp32 A/B split=1"
SSD_IMAGE_SPLIT=1 timeout 600 python tests/micro/imgblock_ab.py 64 2>&1 | tail -45 > $OUT/ab_split1.txt
for wl in 0 1 3; do
  echo "== bf16 WL=$wl"
  SSD_IMAGE_WL=$wl timeout 600 python bench.py --dtype bf16 --layers --steps 30 --no-h2d > $OUT/bf16_wl$wl.json 2> $OUT/bf16_wl$wl.txt
done
grep -h "fuse_image\|total" $OUT/ab_split0.txt $OUT/ab_split1.txt
for wl in 0 1 3; do python - <<PY
import json
d=json.loads(open("$OUT/bf16_wl$wl.json").read().strip().splitlines()[-1]); print("bf16 WL=$wl", d["value"], d["ms_per_step"])
PY
grep "_fused" $OUT/bf16_wl$wl.txt | awk '{print $1,$3,$4}' | tr '\n' ';'; echo; done
