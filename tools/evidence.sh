#!/bin/bash
# Evidence run of a round on one MI355X lease: tools/evidence.sh <tag>   (through gpurun, from the repo root)
# bench lines of every BASELINE configuration that fits one GPU (fp32 and the bf16 mode), per-layer tables, forced
# single-rank collectives, rocprofv3 kernel stats, PMC traffic + SQ counters of the headline configuration.
# Summaries land in gpurun_out/<tag>/; profiles/adopt.sh copies them into profiles/.
TAG=${1:-r5}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > $OUT/bench_mbv2_b64.json 2> $OUT/bench_mbv2_b64.err
python bench.py --lanes 1 --no-overlap --no-other-leg --no-h2d --no-cpu-baseline --layers > $OUT/bench_mbv2_b64_uncontended.json 2> $OUT/bench_mbv2_b64_layers.txt
python bench.py --backbone vgg16 --no-cpu-baseline --layers > $OUT/bench_vgg16_b32.json 2> $OUT/bench_vgg16_b32_layers.txt
python bench.py --img-size 512 --batch 16 --no-cpu-baseline > $OUT/bench_mbv2_512_b16.json 2>/dev/null
python bench.py --train > $OUT/bench_train_mbv2_b32.json 2>/dev/null
python bench.py --train --backbone vgg16 --batch 16 --no-cpu-baseline > $OUT/bench_train_vgg16_b16.json 2>/dev/null
# the bf16 mode (BASELINE configs[3] / [4] per-GPU shards, and configs[1] / [2] shapes for reference)
python bench.py --dtype bf16 --img-size 512 --batch 16 --no-cpu-baseline > $OUT/bench_bf16_mbv2_512_b16.json 2>/dev/null
python bench.py --dtype bf16 --img-size 512 --batch 16 --lanes 1 --no-overlap --no-other-leg --no-h2d --no-cpu-baseline --layers > $OUT/bench_bf16_mbv2_512_b16_uncontended.json 2> $OUT/bench_bf16_mbv2_512_b16_layers.txt
python bench.py --dtype bf16 --train --no-cpu-baseline > $OUT/bench_bf16_train_mbv2_b32.json 2>/dev/null
python bench.py --dtype bf16 --no-cpu-baseline > $OUT/bench_bf16_mbv2_b64.json 2>/dev/null
python bench.py --dtype bf16 --lanes 1 --no-overlap --no-other-leg --no-h2d --no-cpu-baseline --layers > $OUT/bench_bf16_mbv2_b64_uncontended.json 2> $OUT/bench_bf16_mbv2_b64_layers.txt
python bench.py --dtype bf16 --backbone vgg16 --no-cpu-baseline > $OUT/bench_bf16_vgg16_b32.json 2>/dev/null
# the N > 1 code paths at world size 1 (RCCL communicator alive under GPU_MAX_HW_QUEUES=3; bucketed gradient exchange)
python bench.py --no-cpu-baseline --force-dist > $OUT/bench_forcedist_mbv2_b64.json 2> $OUT/bench_forcedist_mbv2_b64.err
python bench.py --train --no-cpu-baseline --force-dist > $OUT/bench_forcedist_train_mbv2_b32.json 2> $OUT/bench_forcedist_train_mbv2_b32.err
for b in 1 16 32 128 256; do python bench.py --batch $b --no-cpu-baseline --no-h2d 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=%-4d %8.0f img/s %.4f ms/step (three in flight) | %8.0f img/s %.4f ms (one at a time)' % ($b, r['value'], r['ms_per_step'], r['other_mode']['images_per_sec'], r['other_mode']['ms_per_step']))"; done > $OUT/batch_sweep.txt 2>&1
profiles/collect.sh ${TAG}_prof > $OUT/collect.log 2>&1
profiles/collect_sq.sh ${TAG}_sq > $OUT/collect_sq.log 2>&1
BENCH_ARGS="--backbone vgg16" profiles/collect.sh ${TAG}_prof_vgg > $OUT/collect_vgg.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_trace -o t -- python bench.py --train --steps 8 --warmup 4 --repeats 1 --no-cpu-baseline > $OUT/train_trace.log 2>&1
for f in $OUT/bench_*.json; do echo "$f: $(cut -c1-220 $f)"; done
cat $OUT/batch_sweep.txt
