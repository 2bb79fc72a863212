export SSD_HIP_TUNE_CACHE=/tmp/tc
echo "gate=1 wait"; python tests/micro/lanes_now.py 2>&1 | tail -2
echo "gate=1 nowait"; SSD_DBG_NOWAIT=1 python tests/micro/lanes_now.py 2>&1 | tail -2
echo "gate=0 nowait"; SSD_HIP_LANE_GATE=0 SSD_DBG_NOWAIT=1 python tests/micro/lanes_now.py 2>&1 | tail -2
