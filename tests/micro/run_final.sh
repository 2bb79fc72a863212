# round-2 evidence run: benches of every config + rocprofv3 passes (summaries are copied to profiles/)
TAG=${1:-r2b}
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash profiles/collect.sh ${TAG}_prof > gpurun_out/${TAG}_collect.log 2>&1
cp gpurun_out/${TAG}_prof/traffic.json profiles/traffic_mobilenet_v2_b64.json
python bench.py --layers > gpurun_out/$TAG/bench_mbv2_b64.json 2> gpurun_out/$TAG/bench_mbv2_b64_layers.txt
python bench.py --backbone vgg16 --layers > gpurun_out/$TAG/bench_vgg16_b32.json 2> gpurun_out/$TAG/bench_vgg16_b32_layers.txt
python bench.py --img-size 512 --batch 16 --no-cpu-baseline > gpurun_out/$TAG/bench_mbv2_512_b16.json 2>/dev/null
python bench.py --train --steps 10 --warmup 3 > gpurun_out/$TAG/bench_train_mbv2_b32.json 2>/dev/null
python bench.py --train --backbone vgg16 --batch 16 --steps 5 --warmup 2 > gpurun_out/$TAG/bench_train_vgg16_b16.json 2>/dev/null
for b in 1 16 32 128 256; do python bench.py --batch $b --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('B=%d %.0f img/s %.3f ms' % ($b, r['value'], r['ms_per_step']))"; done > gpurun_out/$TAG/batch_sweep.txt
python tests/bench_decoder.py > gpurun_out/$TAG/bench_decoder.txt 2>&1
bash profiles/collect_sq.sh ${TAG}_sq > gpurun_out/${TAG}_sq.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/train_trace -o t -- python bench.py --train --steps 5 --warmup 2 > gpurun_out/$TAG/train_trace.log 2>&1
find gpurun_out -name '*kernel_trace.csv' -delete; find gpurun_out -name '*counter_collection.csv' -size +4M -delete
for f in gpurun_out/$TAG/*.json; do echo $f; cut -c1-260 $f; done
cat gpurun_out/$TAG/batch_sweep.txt
