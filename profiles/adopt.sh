#!/bin/bash
# copy the summaries of a tools/evidence.sh run (gpurun_out/<TAG>*) into profiles/ under <TAG>_*
TAG=${1:?tag}
cd "$(dirname "$0")/.."
for f in gpurun_out/$TAG/bench_*.json gpurun_out/$TAG/bench_*_layers.txt gpurun_out/$TAG/batch_sweep.txt; do
    [ -f "$f" ] && cp "$f" profiles/${TAG}_$(basename $f)
done
for f in gpurun_out/$TAG/bench_forcedist_*.err; do [ -f "$f" ] && cp "$f" profiles/${TAG}_$(basename $f .err)_stderr.txt; done
cp gpurun_out/${TAG}_prof/summary.txt profiles/${TAG}_mbv2_b64_rocprofv3_summary.txt
find gpurun_out/${TAG}_prof/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} profiles/${TAG}_mbv2_b64_kernel_stats.csv
cp gpurun_out/${TAG}_sq/sq_summary.txt profiles/${TAG}_mbv2_b64_sq_counters.txt 2>/dev/null
cp gpurun_out/${TAG}_prof_vgg/summary.txt profiles/${TAG}_vgg16_b32_rocprofv3_summary.txt 2>/dev/null
find gpurun_out/${TAG}_prof_vgg/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} profiles/${TAG}_vgg16_b32_kernel_stats.csv
find gpurun_out/$TAG/train_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} profiles/${TAG}_train_mbv2_b32_kernel_stats.csv
cp gpurun_out/${TAG}_prof/traffic.json profiles/traffic_mobilenet_v2_b64.json
cp gpurun_out/${TAG}_prof_vgg/traffic.json profiles/traffic_vgg16_b32.json 2>/dev/null
python - <<'PY'
import json, bench
t = json.load(open("profiles/traffic_mobilenet_v2_b64.json"))
print("traffic hash", t.get("kernel_sources_sha16"), "current", bench.kernel_sources_sha16())
PY
