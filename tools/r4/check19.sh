#!/bin/bash
OUT=gpurun_out/r4s
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_HIP_IGNORE_SHIPPED=1
timeout 1800 python -m pytest tests/test_conv_gpu.py tests/test_bf16_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -5
