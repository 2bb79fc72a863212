#!/bin/bash
# round 5, first lease: hardware probes + the new tests + a baseline bench line at the unchanged kernels
OUT=gpurun_out/r5a; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
{
echo "== lds dma probe"; timeout 60 tests/micro/bin/ldsdma
echo "== graph fork repro"
for q in 2 3 4; do for s in 0 1 2 3; do for p in 0 1; do
  echo "-- queues $q sides $s prio $p"; GPU_MAX_HW_QUEUES=$q timeout 60 tests/micro/bin/gfq $s $p 2>&1 | tail -2; echo "rc=$?"
done; done; done
} > $OUT/probes.log 2>&1
tail -60 $OUT/probes.log
( time timeout 1500 python -m pytest tests -m gpu -q -rA -x -k "default_serving or trained_like" ) > $OUT/newtests.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR|trained-like|bf16 training step on" $OUT/newtests.log | cut -c1-400
( time python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -2 $OUT/bench.err; cut -c1-400 $OUT/bench.json
