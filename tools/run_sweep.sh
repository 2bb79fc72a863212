for b in 1 16 32 128 256; do python bench.py --batch $b --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=%-4d %8.0f img/s %.4f ms/step (two in flight, used=%s) | %8.0f img/s %.4f ms (one at a time)' % ($b, r['value'], r['ms_per_step'], r['config']['lane_calibration']['two_lanes_used'], r['other_mode']['images_per_sec'], r['other_mode']['ms_per_step']))"; done
