#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
FS_IGEMM2=0 bash tools/prof_step.sh c3 5 r04g_c3_old | head -24 | cut -c1-200
FS_IGEMM2=1 bash tools/prof_step.sh c3 5 r04g_c3_new | head -24 | cut -c1-200
