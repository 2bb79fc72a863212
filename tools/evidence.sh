#!/bin/bash
# Evidence run of a round on one MI355X lease: tools/evidence.sh <tag>   (through gpurun, from the repo root)
# bench lines of every BASELINE configuration that fits one GPU, per-layer tables, rocprofv3 kernel stats,
# PMC traffic + SQ counters of the headline configuration.  Summaries land in gpurun_out/<tag>/; copy into profiles/.
TAG=${1:-r3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > $OUT/bench_mbv2_b64.json 2> $OUT/bench_mbv2_b64.err
python bench.py --lanes 1 --no-overlap --no-other-leg --no-cpu-baseline --layers > $OUT/bench_mbv2_b64_uncontended.json 2> $OUT/bench_mbv2_b64_layers.txt
python bench.py --backbone vgg16 --no-cpu-baseline --layers > $OUT/bench_vgg16_b32.json 2> $OUT/bench_vgg16_b32_layers.txt
python bench.py --img-size 512 --batch 16 --no-cpu-baseline > $OUT/bench_mbv2_512_b16.json 2>/dev/null
python bench.py --train > $OUT/bench_train_mbv2_b32.json 2>/dev/null
python bench.py --train --backbone vgg16 --batch 16 --no-cpu-baseline > $OUT/bench_train_vgg16_b16.json 2>/dev/null
for b in 1 16 32 128 256; do python bench.py --batch $b --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=%-4d %8.0f img/s %.4f ms/step (two in flight) | %8.0f img/s %.4f ms (one at a time)' % ($b, r['value'], r['ms_per_step'], r['other_mode']['images_per_sec'], r['other_mode']['ms_per_step']))"; done > $OUT/batch_sweep.txt 2>&1
profiles/collect.sh ${TAG}_prof > $OUT/collect.log 2>&1
profiles/collect_sq.sh ${TAG}_sq > $OUT/collect_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_trace -o t -- python bench.py --train --steps 6 --warmup 2 --no-cpu-baseline > $OUT/train_trace.log 2>&1
for f in $OUT/bench_*.json; do echo "$f: $(cut -c1-260 $f)"; done
cat $OUT/batch_sweep.txt
