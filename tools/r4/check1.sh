#!/bin/bash
# round-4 first GPU check: new tests, default bench line, forced single-rank collectives (inference + training)
OUT=gpurun_out/r4a
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -x -k "nonfinite or accurate" -rA 2>&1 | tail -15 ) > $OUT/newtests.log 2>&1
( timeout 900 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
( timeout 600 python bench.py --no-cpu-baseline --force-dist ) > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err
( timeout 600 python bench.py --train --no-cpu-baseline ) > $OUT/bench_train.json 2> $OUT/bench_train.err
( timeout 600 python bench.py --train --no-cpu-baseline --force-dist ) > $OUT/bench_train_forcedist.json 2> $OUT/bench_train_forcedist.err
for f in $OUT/*.json; do echo "== $f"; cut -c1-700 $f; done
tail -5 $OUT/*.err
cat $OUT/newtests.log
