#!/bin/bash
# Diagnostic builds of the conv kernel with one main-loop phase removed (see SSD_CONV_ABLATE in
# ssd_conv.hip): build/ablate/libssd_hip_ab<bits>.so for tests/micro/conv_ablate.py.
set -e
cd "$(dirname "$0")"
bash build.sh
mkdir -p build/ablate
for n in "$@"; do
  (
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSSD_CONV_ABLATE=$n -c ssd_conv.hip -o build/ablate/conv_$n.o
    objs=$(ls build/*.o | grep -v ssd_conv.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ablate/libssd_hip_ab$n.so build/ablate/conv_$n.o $objs
    echo "built ab$n"
  ) &
done
wait
