#!/bin/bash
OUT=gpurun_out/r4m
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_HIP_WARN_STALE_TABLE=0
( time timeout 1500 python -m pytest tests/test_train.py tests/test_loss.py tests/test_bf16_gpu.py -m gpu -q -x ) > $OUT/t.log 2>&1
grep -E "^E  |passed|failed|real" $OUT/t.log | head
