#!/bin/bash
# Diagnostic builds of the split-bf16 conv kernel with phases removed (SSD_C3_ABLATE in ssd_conv_mfma.h):
# build/ablate/libssd_hip_c3ab<bits>.so for tests/micro/conv3_ablate.py.
set -e
cd "$(dirname "$0")"
bash build.sh
mkdir -p build/ablate
for n in "$@"; do
  (
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Ibuild -DSSD_C3_ABLATE=$n -c ssd_conv3.hip -o build/ablate/conv3_$n.o
    objs=$(ls build/*.o | grep -v ssd_conv3.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ablate/libssd_hip_c3ab$n.so build/ablate/conv3_$n.o $objs
    echo "built c3ab$n"
  ) &
done
wait
