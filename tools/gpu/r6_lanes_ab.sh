cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { python bench.py --steps 20 --warmup 5 --other-configs off --no-cpu-baseline --no-h2d --no-other-leg "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value']), round(r['ms_per_step'],4))"; }
echo -n "lanes 3 / 3 queues: "; run --lanes 3
echo -n "lanes 4 / 4 queues: "; GPU_MAX_HW_QUEUES=4 run --lanes 4
echo -n "lanes 4 / 4 queues, groups cap 1: "; GPU_MAX_HW_QUEUES=4 SSD_IMAGE_GROUPS_CAP=1 run --lanes 4
echo -n "lanes 4 / 3 queues: "; GPU_MAX_HW_QUEUES=3 run --lanes 4
echo -n "lanes 2 / 2 queues: "; run --lanes 2
