#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 600 python -W ignore -m pytest tests/test_program_group_gpu.py tests/test_train_steps_gpu.py -q -x --timeout 300 > $O/r04i_steps.log 2>&1; tail -4 $O/r04i_steps.log | cut -c1-220
for g in 0 1; do
  FS_GROUP_PROGRAMS=$g timeout 200 python -W ignore tools/step_time.py c3 10 2>&1 | grep STEP_TIME | sed "s/^/group=$g /"
  FS_GROUP_PROGRAMS=$g timeout 200 python -W ignore tools/step_time.py c3 10 fp32 2>&1 | grep STEP_TIME | sed "s/^/group=$g /"
  FS_GROUP_PROGRAMS=$g timeout 300 python -W ignore tools/step_time.py c5 6 2>&1 | grep STEP_TIME | sed "s/^/group=$g /"
done
