#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
t() { echo "== $*"; env "$@" timeout 300 python -W ignore tools/debug_group_nan.py graph 2>&1 | grep -E "^step|Error|Segm" | awk '{print $2, $4, $6}' | tr '\n' ';'; echo; }
t FS_GROUP_CAPTURE=2
t FS_GROUP_CAPTURE=2 FS_LAYER_LANES=2
