export SSD_HIP_TUNE_CACHE=/tmp/tc
for p in 0 1 2 0 1 2; do echo "SSD_TAIL_PRIO=$p"; SSD_TAIL_PRIO=$p python tests/micro/tail_ab.py 64 2>&1 | grep "ms/step"; done
SSD_TAIL_PRIO=1 python tests/micro/tail_ab.py 64 2>&1 | tail -16
