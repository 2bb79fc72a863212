cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_HIP_IGNORE_SHIPPED=1
SSD_HIP_USE_GRAPH=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_trace -o t -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --lanes 1 > gpurun_out/tl_bench.log 2>&1
python tests/micro/timeline.py gpurun_out/tl_trace > gpurun_out/r3b_timeline_direct.txt
rm -rf gpurun_out/tl_trace
tail -70 gpurun_out/r3b_timeline_direct.txt | cut -c1-110
