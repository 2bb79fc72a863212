#!/bin/bash
# round 6: full GPU suite, then re-measure every shipped kernel-choice table at the current build
TAG=${1:-r6s}
bash tools/gpu/suite.sh $TAG
OUT=gpurun_out/tables_r6
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time SSD_HIP_IGNORE_SHIPPED=1 SSD_HIP_WARN_STALE_TABLE=0 python tools/make_tuning_tables.py --out $OUT --repeats 3 ) > $OUT/log.txt 2>&1
tail -32 $OUT/log.txt
