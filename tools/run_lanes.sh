for i in 1 2 3; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.0f img/s  %.4f ms/step | other: %.4f ms (%s) | lanes: %s' % (r['value'], r['ms_per_step'], r['other_mode']['ms_per_step'], r['other_mode']['mode'][:18], r['config']['lane_calibration']))"; done
python bench.py --lanes 2 --no-cpu-baseline 2>/dev/null | cut -c1-200
python bench.py --lanes 1 --no-cpu-baseline 2>/dev/null | cut -c1-330
python -m pytest tests/test_fullsize_gpu.py -q -x -k "two_lanes or bench_line" 2>&1 | tail -3
