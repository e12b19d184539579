#!/bin/bash
# round 3, fourth GPU call: whole GPU suite with the 2-stage igemm / 128-pixel wgrad chunks / runtime deterministic mode, micro-benchmarks,
# step timings, LUT regeneration, default bench line
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
mkdir -p $O
rm -f $O/parity_metrics.json
timeout 1500 python -m pytest tests -m gpu -q > $O/r03d_gpu_tests.log 2>&1; tail -8 $O/r03d_gpu_tests.log
timeout 200 python tools/wgrad_micro.py 2>&1 | tail -11
timeout 300 python tools/step_time.py c3 10 2>&1 | grep STEP_TIME
timeout 300 python tools/step_time.py c3 8 fp32 2>&1 | grep STEP_TIME
timeout 300 python tools/step_time.py c5 6 2>&1 | grep STEP_TIME
timeout 300 python tools/step_time.py c4 10 2>&1 | grep STEP_TIME
timeout 300 python tools/step_time.py c4 8 fp32 2>&1 | grep STEP_TIME
timeout 600 python -m fasterseg_amd.latency_lookup_table --quick --out $O/r03_lut_mi355x_bf16.npy > $O/r03d_lut.log 2>&1; tail -2 $O/r03d_lut.log
timeout 1200 python bench.py > $O/r03d_bench.json 2> $O/r03d_bench.err; tail -c 400 $O/r03d_bench.err
python tools/extract_bench.py $O/r03d_bench.json
