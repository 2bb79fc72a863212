#!/bin/bash
# full GPU suite on a fresh lease (no tuning cache: shipped tables + on-device autotune for what they lack)
TAG=${1:-r4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 2700 python -m pytest tests -m gpu -q -rA --durations=15 ) > $OUT/gputest.log 2>&1
echo "pytest rc=$?" >> $OUT/gputest.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $OUT/gputest.log | tail -20
grep -E "bf16 vs fp32|yardstick|bf16 tiles|bf16 training|borderline" $OUT/gputest.log | cut -c1-400
