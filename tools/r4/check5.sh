#!/bin/bash
OUT=gpurun_out/r4f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( timeout 1500 python -m pytest tests/test_bbox_gpu.py tests/test_fullsize_gpu.py tests/test_conv_gpu.py -m gpu -q -x 2>&1 | tail -8 ) > $OUT/tests.log 2>&1
cat $OUT/tests.log
for fs in 1 0; do
  SSD_FUSE_SOFTMAX=$fs python bench.py --lanes 1 --no-other-leg --no-cpu-baseline > $OUT/bench_l1_fs$fs.json 2>/dev/null
  SSD_FUSE_SOFTMAX=$fs python bench.py --batch 1 --lanes 1 --no-other-leg --no-cpu-baseline > $OUT/bench_b1_fs$fs.json 2>/dev/null
  SSD_FUSE_SOFTMAX=$fs python bench.py --no-cpu-baseline > $OUT/bench_l3_fs$fs.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4f/bench_*.json")):
    r=json.loads(open(f).read().strip().splitlines()[-1])
    k=r.get("gpu_ms_per_step_by_kind",{})
    print(f, "%.4f ms/step %.0f img/s"%(r["ms_per_step"], r["value"]), "softmax %.4f nms %.4f"%(k.get("softmax",0),k.get("nms",0)))
PY
