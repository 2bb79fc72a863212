cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6b
export SSD_HIP_WARN_STALE_TABLE=0
for v in ${VARIANTS:-0 1}; do SSD_IMAGE2_VARIANT=$v timeout 300 python tests/micro/imgblock2_ab.py ${1:-64} ${2:-fp32} 2>&1 | grep -v Warn | tee -a gpurun_out/r6b/v2ab.log | grep -E "variant|fused|total|max"; done
