#!/bin/bash
OUT=gpurun_out/r4c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_bf16_gpu.py -m gpu -q -rA -k "forward and mobilenet" > $OUT/bf16tests.log 2>&1
grep -n "bf16 vs fp32\|^E " $OUT/bf16tests.log | head -40
