#!/bin/bash
# round 3, third GPU call: deterministic reductions with per-access coherence (tests, micro-benchmark, step timing A/B), parity tests
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_bn_group_gpu.py -q -x -k "wgrad or dgrad or bn" > $O/r03c_kernels.log 2>&1; tail -3 $O/r03c_kernels.log
timeout 200 python tools/wgrad_micro.py 2>&1 | tail -12
for cfg in "" "FS_WGRAD_ATOMICS=1" "FS_BN_EPILOGUE_STATS=1" "FS_WGRAD_ATOMICS=1 FS_BN_EPILOGUE_STATS=1 FS_BN_ATOMICS=1"; do
  echo "cfg: $cfg"; env $cfg FS_EAGER_LANES=4 timeout 300 python tools/step_time.py c3 10 2>&1 | grep STEP_TIME
done
env FS_EAGER_LANES=4 timeout 300 python tools/step_time.py c4 10 2>&1 | grep STEP_TIME
env FS_WGRAD_ATOMICS=1 FS_BN_EPILOGUE_STATS=1 FS_BN_ATOMICS=1 FS_EAGER_LANES=4 timeout 300 python tools/step_time.py c4 10 2>&1 | grep STEP_TIME
rm -f $O/parity_metrics.json
timeout 900 python -m pytest tests/test_train_parity_gpu.py -q > $O/r03c_parity.log 2>&1; tail -5 $O/r03c_parity.log
timeout 900 python -m pytest tests/test_train_steps_gpu.py tests/test_parallel_gpu.py tests/test_latency_lut.py -q > $O/r03c_steps.log 2>&1; tail -5 $O/r03c_steps.log
