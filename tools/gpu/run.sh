#!/bin/bash
# One parameterised GPU check instead of a script per experiment.  Through gpurun, from the repo root:
#   tools/gpu/run.sh <tag> <what> [args...]      logs under gpurun_out/<tag>/
#     bench   [bench.py args]     one bench line (+ stderr) and its headline numbers
#     layers  [bench.py args]     the uncontended per-layer table (--lanes 1 --no-overlap --layers)
#     pytest  [pytest args]       pytest tests -m gpu <args>
#     micro   <script> [args]     python tests/micro/<script> <args>
#     smoke                       __graft_entry__.smoke()
TAG=${1:?tag}; WHAT=${2:?what}; shift 2
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
case $WHAT in
  bench)
    n=$(ls "$OUT"/bench_*.json 2>/dev/null | wc -l)
    python bench.py "$@" > "$OUT/bench_$n.json" 2> "$OUT/bench_$n.err"; tail -3 "$OUT/bench_$n.err"
    python - "$OUT/bench_$n.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
o = r.get("other_mode") or {}
print("%s: %.0f %s, %.4f ms/step | other: %s %.4f ms | roofline frac %s" % (
    r["config"]["workload"][:60], r["value"], r["unit"], r["ms_per_step"], o.get("mode", "-")[:20], o.get("ms_per_step", 0.0),
    (r.get("roofline") or {}).get("frac")))
PY
    ;;
  layers)
    n=$(ls "$OUT"/layers_*.txt 2>/dev/null | wc -l)
    python bench.py --lanes 1 --no-overlap --no-other-leg --no-h2d --no-cpu-baseline --layers "$@" > "$OUT/layers_$n.json" 2> "$OUT/layers_$n.txt"
    cat "$OUT/layers_$n.txt" | tail -90
    ;;
  pytest) ( time timeout 2700 python -m pytest tests -m gpu -q -rA --durations=15 "$@" ) > "$OUT/gputest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/gputest.log"
    grep -E "passed|failed|^FAILED|^ERROR|rc=" "$OUT/gputest.log" | tail -20 ;;
  micro) s=$1; shift; timeout 1200 python "tests/micro/$s" "$@" 2>&1 | grep -v Warn | tee "$OUT/micro_${s%.*}.log" | tail -60 ;;
  smoke) python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee "$OUT/smoke.log" | tail -5 ;;
  *) echo "unknown: $WHAT" >&2; exit 2 ;;
esac
