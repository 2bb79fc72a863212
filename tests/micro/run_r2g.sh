mkdir -p gpurun_out/r2g
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_train.py -m gpu -q 2>&1 | tail -3
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2g/train_trace -o t -- python bench.py --train --steps 5 --warmup 2 > gpurun_out/r2g/train_trace.log 2>&1
tail -2 gpurun_out/r2g/train_trace.log | cut -c1-400
f=$(find gpurun_out/r2g/train_trace -name "*kernel_stats.csv" | head -1)
head -16 "$f" | cut -c1-150
