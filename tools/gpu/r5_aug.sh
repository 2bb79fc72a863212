#!/bin/bash
OUT=gpurun_out/r5f; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python -m pytest tests/test_augment.py tests/test_convdma_gpu.py -m gpu -q -rA ) > $OUT/tests.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR|Error|assert|dmab tiles" $OUT/tests.log | head -30
