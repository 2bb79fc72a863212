#!/bin/bash
# Profiling build of the split-bf16 conv tiles (-DSSD_C3_PROF): tests/micro/bin/libssd_hip_c3prof.so for tests/micro/conv3_prof.py
set -e
cd "$(dirname "$0")/../../tf-ssd_amd/csrc"
bash build.sh
mkdir -p ../../tests/micro/bin build/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Ibuild -DSSD_C3_PROF=1 -c ssd_conv3.hip -o build/var/conv3_prof.o
objs=$(ls build/*.o | grep -v ssd_conv3.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tests/micro/bin/libssd_hip_c3prof.so build/var/conv3_prof.o $objs
echo "built c3prof"
