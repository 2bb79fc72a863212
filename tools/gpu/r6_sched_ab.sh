cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_HIP_WARN_STALE_TABLE=0
for lib in "" max-ilp max-memory-clause; do
  echo "== scheduler strategy: ${lib:-default}"
  if [ -n "$lib" ]; then export SSD_HIP_LIBRARY=$PWD/tests/micro/bin/libssd_hip_$lib.so; fi
  python tests/micro/imgblock2_ab.py 64 fp32 2>&1 | grep -E "fused|total"
done
