#!/bin/bash
# round 3, first GPU call: new parity tests, eager-lane timing, the reworked default bench line
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
mkdir -p $O
rm -f $O/parity_metrics.json
timeout 900 python -m pytest tests/test_train_parity_gpu.py -q -x > $O/r03a_parity.log 2>&1; tail -3 $O/r03a_parity.log
timeout 900 python -m pytest tests/test_train_steps_gpu.py tests/test_supernet.py tests/test_parallel_gpu.py tests/test_eval_path.py tests/test_engine_gpu.py -q -m gpu > $O/r03a_tests.log 2>&1; tail -3 $O/r03a_tests.log
for lanes in 1 4 6 10; do FS_EAGER_LANES=$lanes timeout 300 python tools/step_time.py c3 10 2>&1 | grep STEP_TIME; done
FS_EAGER_LANES=1 timeout 300 python tools/step_time.py c5 6 2>&1 | grep STEP_TIME
timeout 300 python tools/step_time.py c5 6 2>&1 | grep STEP_TIME
timeout 1200 python bench.py > $O/r03a_bench.json 2> $O/r03a_bench.err; tail -c 600 $O/r03a_bench.err
python tools/extract_bench.py $O/r03a_bench.json
