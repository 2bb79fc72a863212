#!/bin/bash
OUT=gpurun_out/r4q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for sp in 0 1; do echo "== split=$sp"; SSD_IMAGE_SPLIT=$sp timeout 600 python tests/micro/imgblock_prof.py 64 2>&1 | grep -v Warn | tail -12; done
