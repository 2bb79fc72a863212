#!/bin/bash
# round 6: full GPU suite + the driver's bench command + the uncontended per-layer table, one lease
TAG=${1:-r6a}
bash tools/gpu/suite.sh $TAG
bash tools/gpu/run.sh $TAG bench --gpus 1 --steps 20 --warmup 5
bash tools/gpu/run.sh $TAG layers > /dev/null
tail -75 gpurun_out/$TAG/layers_0.txt
