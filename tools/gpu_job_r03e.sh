#!/bin/bash
# round 3, fifth GPU call: 64x64 two-stage igemm configuration - kernel tests, sweep on the fused shapes, step timing A/B
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x > $O/r03e_kernels.log 2>&1; tail -3 $O/r03e_kernels.log
FS_SWEEP_ONLY_R3=1 FS_SWEEP_DTYPE=bf16 timeout 300 python tools/conv_sweep.py 2>&1 | tail -15
for w in 0 96 48; do echo "FS_IGEMM_WIDE=$w"; FS_IGEMM_WIDE=$w timeout 300 python tools/step_time.py c3 10 2>&1 | grep STEP_TIME; done
FS_IGEMM_WIDE=0 timeout 300 python tools/step_time.py c3 8 fp32 2>&1 | grep STEP_TIME
FS_IGEMM_WIDE=96 timeout 300 python tools/step_time.py c3 8 fp32 2>&1 | grep STEP_TIME
FS_IGEMM_WIDE=0 timeout 300 python tools/step_time.py c5 6 2>&1 | grep STEP_TIME
FS_IGEMM_WIDE=96 timeout 300 python tools/step_time.py c5 6 2>&1 | grep STEP_TIME
timeout 600 python -m pytest tests/test_train_steps_gpu.py tests/test_latency_lut.py -q > $O/r03e_steps.log 2>&1; tail -3 $O/r03e_steps.log
