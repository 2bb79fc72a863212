#!/bin/bash
# kernel trace of the three-lane region vs the one-at-a-time region: which kernels inflate when the lanes share the chip
OUT=gpurun_out/r4y
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_HIP_WARN_STALE_TABLE=0
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/l3 -o t -- python bench.py --steps 30 --warmup 6 --repeats 1 --no-cpu-baseline --no-h2d --no-other-leg > $OUT/l3.log 2>&1
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/l1 -o t -- python bench.py --steps 30 --warmup 6 --repeats 1 --no-cpu-baseline --no-h2d --no-other-leg --lanes 1 > $OUT/l1.log 2>&1
python tools/gpu/lanes_trace.py $OUT | tee $OUT/summary.txt
rm -rf $OUT/l3 $OUT/l1
