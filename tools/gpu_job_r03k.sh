#!/bin/bash
# round 3, final GPU call: the LDS tap-table gather of the small-tile igemm configurations - kernel tests, sweep, step tests, step time, LUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 200 python -m pytest tests/test_kernels_gpu.py -q -x > $O/r03k_kernels.log 2>&1; tail -2 $O/r03k_kernels.log
FS_SWEEP_ONLY_R3=1 FS_SWEEP_DTYPE=bf16 FS_SWEEP_CFGS=-1,4,5,6 timeout 120 python tools/conv_sweep.py 2>&1 | tail -14
timeout 300 python -m pytest tests/test_train_steps_gpu.py tests/test_ops_gpu.py tests/test_zoom_cell_gpu.py -q > $O/r03k_steps.log 2>&1; tail -2 $O/r03k_steps.log
timeout 200 python tools/step_time.py c3 10 2>&1 | grep STEP_TIME
timeout 200 python -m fasterseg_amd.latency_lookup_table --quick --out $O/r03k_lut_mi355x_bf16.npy > $O/r03k_lut.log 2>&1; tail -1 $O/r03k_lut.log
