#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python tools/stream_overlap.py 2>&1 | tail -6
