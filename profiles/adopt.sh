#!/bin/bash
# copy the summaries of a tests/micro/run_final.sh evidence run (gpurun_out/<TAG>*) into profiles/ under <TAG>_*
TAG=${1:?tag}
cd "$(dirname "$0")/.."
for f in batch_sweep.txt bench_decoder.txt bench_mbv2_512_b16.json bench_mbv2_b64.json bench_mbv2_b64_layers.txt \
         bench_train_mbv2_b32.json bench_train_vgg16_b16.json bench_vgg16_b32.json bench_vgg16_b32_layers.txt; do
    [ -f gpurun_out/$TAG/$f ] && cp gpurun_out/$TAG/$f profiles/${TAG}_$f
done
cp gpurun_out/${TAG}_prof/summary.txt profiles/${TAG}_mbv2_b64_rocprofv3_summary.txt
cp gpurun_out/${TAG}_prof/trace/*kernel_stats.csv profiles/${TAG}_mbv2_b64_kernel_stats.csv 2>/dev/null
cp gpurun_out/${TAG}_sq/sq_summary.txt profiles/${TAG}_mbv2_b64_sq_counters.txt 2>/dev/null
cp gpurun_out/$TAG/train_trace/*kernel_stats.csv profiles/${TAG}_train_mbv2_b32_kernel_stats.csv 2>/dev/null
cp gpurun_out/${TAG}_prof/traffic.json profiles/traffic_mobilenet_v2_b64.json
python - <<'PY'
import json, bench
t = json.load(open("profiles/traffic_mobilenet_v2_b64.json"))
print("traffic hash", t.get("kernel_sources_sha16"), "current", bench.kernel_sources_sha16())
PY
