#!/bin/bash
OUT=gpurun_out/r5j; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for o in "vgg conv4_2" "mbv2 head1" "vgg conv2_2"; do ONLY="$o" timeout 600 python tests/micro/convdma_ab.py 64 v 2>&1 | grep -E "dma3_2x4_4x2|dmab_2x4_4x2|dmab_4x4_4x2|dmab_4x4_2x4|mfma3_2x4_4x2 |^vgg|^mbv2" | grep -E "split 1 |split 2 |^vgg|^mbv2" ; done | tee $OUT/phased.log | cut -c1-200
