#!/bin/bash
# SQ counters (MFMA busy, wait states, LDS conflicts) of the bench kernels; own pass, kernel-trace only.
TAG=${1:-r1sq}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_HIP_TUNE_CACHE=$OUT/tune
# one step at a time, no intra-step side streams, no second timing leg: every kernel runs alone, so the per-kernel
# averages of the trace are uncontended and reproduce the bench line's per-layer hipEvent times
CMD="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --lanes 1 --no-overlap --no-other-leg --no-h2d $BENCH_ARGS"
timeout 600 $CMD > $OUT/bench_plain.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $OUT/sq -o s -- $CMD > $OUT/bench_sq.log 2>&1
tail -3 $OUT/bench_sq.log | cut -c1-200
python - <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out=os.environ.get("OUTDIR","gpurun_out/%s" % (sys.argv[1] if len(sys.argv)>1 else "r1sq"))
PY
python profiles/summarize_sq.py $OUT > $OUT/sq_summary.txt 2>&1
cat $OUT/sq_summary.txt | head -60
