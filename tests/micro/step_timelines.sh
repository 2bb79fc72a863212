# Kernel timelines of one bench step in both launch modes (rocprofv3 --kernel-trace) -> gpurun_out/<tag>_timeline_{direct,replay}.txt
TAG=${1:-tl}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mode in 0 1; do
  name=$([ $mode = 0 ] && echo direct || echo replay)
  SSD_HIP_USE_GRAPH=$mode rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_trace -o t -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python tests/micro/timeline.py gpurun_out/${TAG}_trace > gpurun_out/${TAG}_timeline_${name}.txt
  rm -rf gpurun_out/${TAG}_trace
  tail -1 gpurun_out/${TAG}_timeline_${name}.txt
done
