#!/bin/bash
# round-4 bf16 mode first GPU check
OUT=gpurun_out/r4b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( timeout 1500 python -m pytest tests/test_bf16_gpu.py -m gpu -q -rA 2>&1 | tail -60 ) > $OUT/bf16tests.log 2>&1
( timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --layers ) > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
( timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --img-size 512 --batch 16 ) > $OUT/bench_bf16_512.json 2> $OUT/bench_bf16_512.err
( timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --backbone vgg16 ) > $OUT/bench_bf16_vgg.json 2> $OUT/bench_bf16_vgg.err
( timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --train ) > $OUT/bench_bf16_train.json 2> $OUT/bench_bf16_train.err
( timeout 600 python bench.py --no-cpu-baseline ) > $OUT/bench_f32.json 2> $OUT/bench_f32.err
cat $OUT/bf16tests.log
for f in $OUT/*.json; do echo "== $f"; cut -c1-400 $f; done
for f in $OUT/*.err; do echo "== $f"; tail -n 4 $f; done
