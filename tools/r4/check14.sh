#!/bin/bash
OUT=gpurun_out/r4o
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_HIP_WARN_STALE_TABLE=0
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_trace -o t -- python bench.py --train --steps 8 --warmup 3 --repeats 1 --no-cpu-baseline > $OUT/train_trace.log 2>&1
find $OUT/train_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/train_kernel_stats.csv
head -40 $OUT/train_kernel_stats.csv | cut -c1-160
