cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_HIP_WARN_STALE_TABLE=0 SSD_HIP_LIBRARY=$PWD/tests/micro/bin/libssd_hip_bandabl.so
for a in 0 1 2 4 7 8 16 32 64 15 31 63 127; do echo -n "ablate $a: "; SSD_FUSED_ABLATE=$a python tests/prof_fused.py 2>&1 | grep -E "block_[12]_fused" | sed 's/_fused: total//; s/ cycles.*//' | tr '\n' ' '; echo; done
