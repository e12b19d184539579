#!/bin/bash
# round 4, closing run: statistics switch + big-map routing checked, default bench, rocprofv3 tables of the final steps
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 400 python -W ignore -m pytest tests/test_conv_unit_gpu.py tests/test_train_parity_gpu.py tests/test_train_steps_gpu.py -m gpu -k 'student or conv_bn_relu' -q --timeout 300 > $O/r04w_tests.log 2>&1; tail -3 $O/r04w_tests.log | cut -c1-250
timeout 900 python bench.py --steps 20 --warmup 5 --detail $O/r04w_bench_detail.json > $O/r04w_bench.json 2> $O/r04w_bench.err; tail -c 400 $O/r04w_bench.json; echo
bash tools/prof_step.sh c4 5 r04w_c4_student_train_bf16 2>&1 | head -1
bash tools/prof_step.sh c3 3 r04w_c3_supernet_pretrain_bf16 2>&1 | head -1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  FS_PROFILE_WARMUP=1 timeout 110 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o run -- python $R/tools/profile_step.py c3 1 > $O/r04w_pmc_c3_$c.log 2>&1
done
F=$(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python $R/tools/pmc_table.py $O/r04_c3_pmc.json f=$F w=$W | head -8
