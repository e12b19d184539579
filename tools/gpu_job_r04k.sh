#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
t() { echo "== $*"; env "$@" timeout 300 python -W ignore -m pytest tests/test_train_steps_gpu.py -q -x --timeout 200 -k test_graphed_supernet_step_equals_eager 2>&1 | grep -E "passed|failed|Segmentation|Error" | head -3; }
t FS_X=1
t FS_RECORD_STREAM=0
t FS_GROUP_PROGRAMS=0
t FS_GROUP_PROGRAMS=0 FS_RECORD_STREAM=0
