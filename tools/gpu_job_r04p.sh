#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
t() { env "$@" timeout 300 python -W ignore tools/step_time.py $W 10 $D 2>&1 | grep STEP_TIME | sed "s/^/$* /"; }
W=c3; D=
t FS_GROUP_CAPTURE=0
t FS_GROUP_CAPTURE=2
t FS_GROUP_PROGRAMS=0
W=c3; D=fp32
t FS_GROUP_CAPTURE=0
t FS_GROUP_CAPTURE=2
W=c5; D=
t FS_GROUP_CAPTURE=0
t FS_GROUP_CAPTURE=2
t FS_GROUP_PROGRAMS=0
