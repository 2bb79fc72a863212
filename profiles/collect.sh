#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the
# repo root):  profiles/collect.sh <tag>
# Pass 1: --kernel-trace --stats (per-kernel durations).  Passes 2/3: PMC counters in their
# own runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with trace domains
# other than --kernel-trace).  Output under gpurun_out/<tag>/; copy the summaries to profiles/.
TAG=${1:-r1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# one step at a time, no intra-step side streams, no second timing leg: every kernel runs alone, so the per-kernel
# averages of the trace are uncontended and reproduce the bench line's per-layer hipEvent times
# BENCH_ARGS: another configuration of the same command (e.g. "--backbone vgg16")
CMD="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --lanes 1 --no-overlap --no-other-leg --no-h2d $BENCH_ARGS"
# first run only fills the tuning cache, so that the profiled runs contain no autotune launches
export SSD_HIP_TUNE_CACHE=$OUT/tune
timeout 600 $CMD > $OUT/bench_plain.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/bench_trace.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- $CMD > $OUT/bench_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- $CMD > $OUT/bench_write.log 2>&1
find $OUT -name "*.csv" | head -20
TRAFFIC_JSON=$OUT/traffic.json python profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
tail -60 $OUT/summary.txt
