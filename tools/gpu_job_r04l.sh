#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
which gdb >/dev/null 2>&1 && echo "gdb present"
AMD_LOG_LEVEL=3 FS_GROUP_PROGRAMS=1 timeout 300 python -m pytest tests/test_train_steps_gpu.py -q -x --timeout 200 -k test_graphed_supernet_step_equals_eager > /tmp/log.txt 2>&1
grep -n "Segmentation" /tmp/log.txt | head -2
grep -v "^$" /tmp/log.txt | grep -B40 "Fatal Python error" | grep -E "hip[A-Z]" | tail -40 | cut -c1-240
