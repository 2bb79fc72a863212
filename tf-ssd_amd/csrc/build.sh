#!/bin/bash
# Builds libssd_hip.so (gfx950 only) in-tree.  Usage: build.sh [-f]
set -e
cd "$(dirname "$0")"
OUT=../libssd_hip.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p build
rm -f build/.failed
need_link=0
compile() {  # src extra-flags
  local src=$1; shift
  local obj=build/${src%.*}.o
  if [ "$FORCE" = 1 ] || [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ common.h -nt "$obj" ] \
     || [ ../../include/ssd_hip.h -nt "$obj" ] || [ ssd_conv.h -nt "$obj" ] || [ ssd_net.h -nt "$obj" ] || [ ssd_conv_mfma.h -nt "$obj" ] || [ ssd_bf16x3.h -nt "$obj" ] || { [ -f "${src%.*}.h" ] && [ "${src%.*}.h" -nt "$obj" ]; }; then
    echo "hipcc $src"
    # translation units compile in parallel (ssd_conv.hip alone instantiates ~80 kernels)
    ( $HIPCC $COMMON "$@" -c "$src" -o "$obj.tmp" && mv "$obj.tmp" "$obj" ) || touch build/.failed &
    need_link=1
  fi
}
[ "$1" = "-f" ] && FORCE=1
# build id = sha256 of the kernel sources (same bytes, same order as bench.py's kernel_sources_sha16; the whole recipe: bench.py build_id_from_sources):
# shipped tuning tables and PMC traffic passes record the build they were measured on
# (+ the public header and the compile flags: either changes the library without touching a kernel source)
BID=$( { LC_ALL=C ls *.hip *.h | LC_ALL=C sort | xargs cat; cat ../../include/ssd_hip.h; echo "$COMMON"; } | sha256sum | cut -c1-16)
echo "#define SSD_BUILD_ID \"$BID\"" > build/build_id.h.new
if cmp -s build/build_id.h.new build/build_id.h; then rm build/build_id.h.new; else mv build/build_id.h.new build/build_id.h; rm -f build/ssd_core.o; fi
# box math: separately-rounded fp32 ops (bit-exact indices vs the oracle)
compile ssd_core.hip
compile ssd_bbox.hip -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
# loss: separately rounded ops too (the hard-negative RANK depends on the per-anchor CE values)
compile ssd_loss.hip -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
compile ssd_data.hip -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
for s in ssd_conv.hip ssd_conv3.hip ssd_convdma.hip ssd_wino.hip ssd_skinny.hip ssd_ops.hip ssd_fused.hip ssd_bandblock.hip ssd_band3.hip ssd_imgblock.hip ssd_imgblock2.hip ssd_dwproj.hip ssd_net.hip ssd_train.hip; do
  [ -f "$s" ] && compile "$s"
done
wait
if [ -f build/.failed ]; then rm -f build/.failed; echo "compilation failed" >&2; exit 1; fi
if [ $need_link = 1 ] || [ ! -f $OUT ]; then
  echo "link $OUT"
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT build/*.o
fi
