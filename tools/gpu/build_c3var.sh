#!/bin/bash
# Experimental builds of the split-bf16 conv tiles: tests/micro/bin/libssd_hip_c3v<N>.so with -DSSD_C3_VARIANT=N
# (tests/micro/bin/ is git-ignored but travels with gpurun).  usage: tools/gpu/build_c3var.sh 1 2 ...
set -e
cd "$(dirname "$0")/../../tf-ssd_amd/csrc"
bash build.sh
mkdir -p ../../tests/micro/bin build/var
for n in "$@"; do
  (
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Ibuild -DSSD_C3_VARIANT=$n -c ssd_conv3.hip -o build/var/conv3_$n.o
    objs=$(ls build/*.o | grep -v ssd_conv3.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tests/micro/bin/libssd_hip_c3v$n.so build/var/conv3_$n.o $objs
    echo "built c3v$n"
  ) &
done
wait
