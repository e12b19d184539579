#!/bin/bash
# round 4, call f: conv_igemm2 under the fitted cost model - kernel tests, operator / train-step tests, step times with and without it
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x --timeout 120 > $O/r04f_kernels.log 2>&1; tail -3 $O/r04f_kernels.log | cut -c1-200
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_train_steps_gpu.py tests/test_zoom_cell_gpu.py tests/test_bn_group_gpu.py -q --timeout 300 > $O/r04f_steps.log 2>&1; tail -5 $O/r04f_steps.log | cut -c1-200
for m in 0 1; do
  FS_IGEMM2=$m timeout 200 python tools/step_time.py c3 10 2>&1 | grep STEP_TIME | sed "s/^/igemm2=$m /"
  FS_IGEMM2=$m timeout 200 python tools/step_time.py c3 10 fp32 2>&1 | grep STEP_TIME | sed "s/^/igemm2=$m /"
  FS_IGEMM2=$m timeout 300 python tools/step_time.py c5 6 2>&1 | grep STEP_TIME | sed "s/^/igemm2=$m /"
done
