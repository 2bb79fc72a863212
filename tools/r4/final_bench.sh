#!/bin/bash
OUT=gpurun_out/r4_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time python bench.py ) > $OUT/bench_mbv2_b64.json 2> $OUT/bench_mbv2_b64.err
tail -4 $OUT/bench_mbv2_b64.err
python -c "
import json
r=json.loads(open('$OUT/bench_mbv2_b64.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline']['traffic'], r['roofline']['frac'], r['config']['kernel_table'])"
python -c "import __graft_entry__ as g; g.smoke()"
