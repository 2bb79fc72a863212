#!/bin/bash
# Full GPU suite + default bench on a fresh MI355X lease; logs under gpurun_out/<tag>_*.
# usage (from the repo root, through gpurun): tools/gpu_suite.sh <tag> [extra pytest args]
tag=${1:-run}; shift
mkdir -p gpurun_out
export SSD_HIP_TUNE_CACHE=${SSD_HIP_TUNE_CACHE-gpurun_out/tables_raw}
[ -z "$SSD_HIP_TUNE_CACHE" ] && unset SSD_HIP_TUNE_CACHE
( time timeout 2400 python -m pytest tests -m gpu -q -rA --durations=25 "$@" ) > gpurun_out/${tag}_gputest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${tag}_gputest.log
tail -5 gpurun_out/${tag}_gputest.log
( time timeout 900 python bench.py ) > gpurun_out/${tag}_bench.log 2>&1
tail -3 gpurun_out/${tag}_bench.log | cut -c1-600
