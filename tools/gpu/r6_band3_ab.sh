cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_HIP_WARN_STALE_TABLE=0
for v in 0 1 0 1; do python tests/micro/band3_dma_ab.py $v 2>&1 | grep -v Warn | grep -E "image_v2|block|stem"; done
