#!/bin/bash
OUT=gpurun_out/r4p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do timeout 600 python -m pytest tests/test_bf16_gpu.py -m gpu -q -rA -k training 2>&1 | grep -E "bf16 training|passed|failed|^E " ; done
