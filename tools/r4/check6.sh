#!/bin/bash
OUT=gpurun_out/r4g
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_fullsize_gpu.py tests/test_bbox_gpu.py tests/test_conv_gpu.py -m gpu -q -x > $OUT/t1.log 2>&1
grep -n "^E  \|Error\|passed\|failed" $OUT/t1.log | head -30
