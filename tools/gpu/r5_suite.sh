#!/bin/bash
TAG=${1:-r5s}; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 2700 python -m pytest tests -m gpu -q -rA --durations=15 ) > $OUT/gputest.log 2>&1
echo "pytest rc=$?" >> $OUT/gputest.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $OUT/gputest.log | tail -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
