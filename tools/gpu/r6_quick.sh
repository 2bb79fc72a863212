#!/bin/bash
# round 6: re-measure ONE table (default mobilenet_v2:300:64), adopt it on the box, headline bench + per-layer table
TAG=${1:-r6q}; SHAPE=${2:-mobilenet_v2:300:64}
OUT=gpurun_out/$TAG
mkdir -p $OUT/tables
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SSD_HIP_IGNORE_SHIPPED=1 SSD_HIP_WARN_STALE_TABLE=0 python tools/make_tuning_tables.py --out $OUT/tables --repeats 1 --shapes $SHAPE 2>&1 | grep -v Warn | tail -3
python tools/make_tuning_tables.py --adopt $OUT/tables | tail -1
grep " image " $OUT/tables/*.tune
export SSD_HIP_WARN_STALE_TABLE=0
bash tools/gpu/run.sh $TAG bench --gpus 1 --steps 20 --warmup 5 --other-configs off --no-cpu-baseline --no-h2d
bash tools/gpu/run.sh $TAG layers > /dev/null
grep -E "fused|heads|Conv_1|extra1" $OUT/layers_0.txt | grep -v " 0.00 GFLOP" 
