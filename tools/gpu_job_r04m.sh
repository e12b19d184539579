#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in one_bucket two_buckets k1 all; do
  timeout 120 python -W ignore tools/group_capture_probe2.py $v > /tmp/o.txt 2>&1; echo "rc=$? $(grep -E 'eager ok|captured|replayed|Segmentation|Error' /tmp/o.txt | grep -v Warning | tr '\n' ' ' | cut -c1-200) <- $v"
done
FS_GROUP_PROGRAMS=0 timeout 120 python -W ignore tools/group_capture_probe2.py all > /tmp/o.txt 2>&1; echo "rc=$? $(grep -E 'eager ok|captured|replayed|Segmentation|Error' /tmp/o.txt | grep -v Warning | tr '\n' ' ' | cut -c1-200) <- all ungrouped"
