cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6b
for v in 0 1; do SSD_IMAGE2_VARIANT=$v SSD_HIP_WARN_STALE_TABLE=0 timeout 300 python tests/micro/imgblock2_ab.py 64 2>&1 | grep -v Warn | tee -a gpurun_out/r6b/v2ab.log | grep -E "variant|fused|total|max"; done
