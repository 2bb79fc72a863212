#!/bin/bash
# net-level: LDS-DMA tiles picked by the on-device autotune (shipped tables ignored), bf16 mode and fp32
OUT=gpurun_out/r5c; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_HIP_IGNORE_SHIPPED=1 SSD_HIP_WARN_STALE_TABLE=0
for cfg in "--dtype bf16" "--dtype bf16 --backbone vgg16" "" "--backbone vgg16"; do
  tag=$(echo "$cfg" | tr -d ' -' ); tag=${tag:-f32mbv2}
  python bench.py $cfg --lanes 1 --no-overlap --no-other-leg --no-h2d --no-cpu-baseline --layers > $OUT/layers_$tag.json 2> $OUT/layers_$tag.txt
  grep -E "dma|conv_heads|Conv_1 |extra1|fc7|conv4_2|conv1_2" $OUT/layers_$tag.txt | head -30 | cut -c1-200
  python - $OUT/layers_$tag.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "%.0f img/s %.4f ms/step" % (r["value"], r["ms_per_step"]), r["gpu_ms_per_step_by_kind"])
PY
done
SSD_HIP_CONV_DMA=0 python bench.py --dtype bf16 --lanes 1 --no-overlap --no-other-leg --no-h2d --no-cpu-baseline > $OUT/nodma_bf16.json 2>/dev/null
python -c "
import json
r=json.loads(open('$OUT/nodma_bf16.json').read().strip().splitlines()[-1]); print('bf16 mbv2 without conv_dma: %.0f img/s %.4f ms' % (r['value'], r['ms_per_step']))"
( time timeout 1500 python -m pytest tests -m gpu -q -x -k "bf16 or forward_parity or full_batch_forward_and_decode" ) > $OUT/tests.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" $OUT/tests.log | head
