mkdir -p gpurun_out/r2f
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python tests/micro/dual_stream.py > gpurun_out/r2f/dual_stream.log 2>&1
tail -6 gpurun_out/r2f/dual_stream.log
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f/train_trace -o t -- python bench.py --train --steps 5 --warmup 2 > gpurun_out/r2f/train_trace.log 2>&1
f=$(find gpurun_out/r2f/train_trace -name "*kernel_stats.csv" | head -1)
head -25 "$f" | cut -c1-160
