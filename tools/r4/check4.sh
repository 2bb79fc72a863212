#!/bin/bash
OUT=gpurun_out/r4d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python tests/micro/bf16_layers.py 4 300 > $OUT/layers.log 2>&1
cat $OUT/layers.log | grep -v Warning | head -150
