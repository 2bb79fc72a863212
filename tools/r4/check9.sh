#!/bin/bash
OUT=gpurun_out/r4j
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python tests/micro/conv3_variants.py 0 $@ > $OUT/variants.txt 2>&1
cat $OUT/variants.txt
