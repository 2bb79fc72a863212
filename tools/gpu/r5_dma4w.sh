#!/bin/bash
# 4-wave large-register-tile LDS-DMA configs against the rest (verbose per-config timings)
OUT=gpurun_out/r5_dma4w; mkdir -p $OUT
for s in "mbv2 head1" "mbv2 head2" "mbv2 Conv_1" "mbv2 extra1_2" "vgg conv4_2" "vgg conv3_2"; do
  ONLY="$s" timeout 600 python tests/micro/convdma_ab.py 64 v 2>&1 | grep -v amdgpu.ids | grep "dma3\|dmab_8x\|dmab_4x8\|dmab_4x7\|mfma3.*TF" > "$OUT/$(echo $s | tr ' ' '_').txt"
  tail -1 "$OUT/$(echo $s | tr ' ' '_').txt" | cut -c1-400
done
