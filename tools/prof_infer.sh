#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q > $O/final_gpu_tests.log 2>&1; tail -1 $O/final_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/final_smoke.log 2>&1; tail -1 $O/final_smoke.log
python bench.py --dump-plan $O/plan_bf16.json > $O/final_bench.json 2> $O/final_bench.err; head -c 300 $O/final_bench.json; echo
python bench.py --dtype fp32 --no-cpu-baseline > $O/final_bench_fp32.json 2>/dev/null; grep -o "\"value\": [0-9.]*" $O/final_bench_fp32.json | head -1
cd /tmp
prof() {  # name, env, steps, args...
  name=$1; shift; envs=$1; shift; steps=$1; shift
  rm -rf /tmp/prof_$name
  env $envs timeout 400 rocprofv3 "$@" -d /tmp/prof_$name -o run -- python $R/bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-roofline > $O/prof_$name.log 2>&1
}
prof kt "X=1" 100 --kernel-trace --stats --output-format csv
cp $(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1) $O/r01_student_infer_bf16_kernel_stats.csv
prof kt1 "FS_ENGINE_LANES=1" 100 --kernel-trace --stats --output-format csv
cp $(find /tmp/prof_kt1 -name "*kernel_stats.csv" | head -1) $O/r01_student_infer_bf16_kernel_stats_serial_graph.csv
prof fetch "X=1" 20 --kernel-trace --pmc FETCH_SIZE --output-format csv
head -6000 $(find /tmp/prof_fetch -name "*counter_collection.csv" | head -1) > $O/r01_student_infer_bf16_pmc_FETCH_SIZE.csv
prof write "X=1" 20 --kernel-trace --pmc WRITE_SIZE --output-format csv
head -6000 $(find /tmp/prof_write -name "*counter_collection.csv" | head -1) > $O/r01_student_infer_bf16_pmc_WRITE_SIZE.csv
cd $R
timeout 300 python bench.py --workload student_train --dtype bf16 --steps 10 --warmup 3 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*"
