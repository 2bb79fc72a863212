#!/bin/bash
OUT=gpurun_out/r5b; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python -m pytest tests/test_convdma_gpu.py -m gpu -q -rA -x ) > $OUT/dmatests.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR|Error|assert" $OUT/dmatests.log | head -30
timeout 1200 python tests/micro/convdma_ab.py 64 > $OUT/convdma_ab.log 2>&1
cat $OUT/convdma_ab.log | cut -c1-420
