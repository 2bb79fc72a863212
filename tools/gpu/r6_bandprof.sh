cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_HIP_WARN_STALE_TABLE=0
python tests/prof_fused.py 2>&1 | grep block_
for a in 0 1 2 4 7; do echo "ablate $a"; SSD_FUSED_ABLATE=$a python tests/prof_fused.py 2>&1 | grep block_ | cut -c1-60; done
