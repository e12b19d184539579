#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
for k in 2 4 8; do
  timeout 120 python tools/group_capture_probe.py $k fwd 2>&1 | grep -E "ok|captured|Error|error|fault" | tr '\n' ' '; echo
  timeout 120 python tools/group_capture_probe.py $k bwd 2>&1 | grep -E "ok|captured|Error|error|fault" | tr '\n' ' '; echo
done
FS_GROUP_PROGRAMS=0 timeout 120 python tools/group_capture_probe.py 4 bwd 2>&1 | grep -E "ok|captured|Error|error|fault" | tr '\n' ' '; echo
