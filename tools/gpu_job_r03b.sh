#!/bin/bash
# round 3, second GPU call: deterministic wgrad, fused MixedOp programs (tests + step timing A/B)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "wgrad or dgrad" > $O/r03b_kernels.log 2>&1; tail -3 $O/r03b_kernels.log
timeout 900 python -m pytest tests/test_train_steps_gpu.py -q -x > $O/r03b_steps.log 2>&1; tail -5 $O/r03b_steps.log
rm -f $O/parity_metrics.json
timeout 900 python -m pytest tests/test_train_parity_gpu.py -q > $O/r03b_parity.log 2>&1; tail -5 $O/r03b_parity.log
timeout 600 python -m pytest tests/test_parallel_gpu.py tests/test_supernet.py tests/test_ops_gpu.py -q > $O/r03b_misc.log 2>&1; tail -5 $O/r03b_misc.log
FS_FUSE_MIXEDOP=0 FS_EAGER_LANES=4 timeout 300 python tools/step_time.py c3 10 2>&1 | grep STEP_TIME
FS_FUSE_MIXEDOP=1 FS_EAGER_LANES=4 timeout 300 python tools/step_time.py c3 10 2>&1 | grep STEP_TIME
FS_FUSE_MIXEDOP=1 FS_EAGER_LANES=4 FS_WGRAD_ATOMICS=1 timeout 300 python tools/step_time.py c3 10 2>&1 | grep STEP_TIME
FS_FUSE_MIXEDOP=1 FS_EAGER_LANES=4 timeout 300 python tools/step_time.py c5 6 2>&1 | grep STEP_TIME
FS_EAGER_LANES=4 timeout 300 python tools/step_time.py c4 10 2>&1 | grep STEP_TIME
