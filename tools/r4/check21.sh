#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python tests/micro/train_launch_bound.py 2>&1 | grep -v Warn | tail -12
