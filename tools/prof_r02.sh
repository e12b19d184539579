#!/bin/bash
# Round-2 evidence on one MI355X (run through gpurun from the repo root): GPU test suite, smoke, the default bench line,
# rocprofv3 summaries of the C2 frame (kernel stats, one-frame timeline, FETCH_SIZE / WRITE_SIZE passes) and the per-step kernel
# tables of C3 / C4 / C5.  Everything lands in gpurun_out/; the summaries are copied to profiles/ by hand afterwards.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
O=$R/gpurun_out
cd $R
what=${1:-all}
if [ $what = all ] || [ $what = tests ]; then
  timeout 900 python -m pytest tests -m gpu -q > $O/r02_gpu_tests.log 2>&1; tail -1 $O/r02_gpu_tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r02_smoke.log 2>&1; tail -1 $O/r02_smoke.log
fi
if [ $what = all ] || [ $what = bench ]; then
  timeout 900 python bench.py > $O/r02_bench_default.json 2> $O/r02_bench_default.err
  python tools/extract_bench.py $O/r02_bench_default.json
fi
if [ $what = all ] || [ $what = c2prof ]; then
  # tune the plan once on the idle device, then replay exactly that plan under the profiler (FS_ENGINE_PLAN)
  export FS_ENGINE_PLAN=$O/r02_c2_plan_choices.json
  rm -f $FS_ENGINE_PLAN.*
  timeout 300 python bench.py --workloads c2 --no-cpu-baseline --no-class-map --dump-plan $O/r02_c2_plan_bf16.json > $O/r02_bench_c2_planned.json 2>/dev/null
  python tools/extract_c2.py $O/r02_bench_c2_planned.json
  cd /tmp
  prof() {  # name, steps, rocprof args...
    name=$1; shift; steps=$1; shift
    rm -rf /tmp/prof_$name
    timeout 400 rocprofv3 "$@" --output-format csv -d /tmp/prof_$name -o run -- python $R/bench.py --workloads c2 --steps $steps --warmup 10 --no-cpu-baseline --no-roofline --no-class-map > $O/r02_prof_$name.log 2>&1
  }
  prof kt 200 --kernel-trace --stats
  cp $(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1) $O/r02_c2_infer_bf16_kernel_stats.csv
  python $R/tools/frame_timeline.py $(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1) $O/r02_c2_infer_bf16_frame_timeline.csv | head -3
  prof fetch 30 --kernel-trace --pmc FETCH_SIZE
  prof write 30 --kernel-trace --pmc WRITE_SIZE
  F=$(find /tmp/prof_fetch -name "*counter_collection.csv" | head -1); W=$(find /tmp/prof_write -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_traffic.py $F $W bf16 $O/r02_pmc_traffic.json
  tail -3000 $F > $O/r02_c2_infer_bf16_pmc_FETCH_SIZE_tail.csv; tail -3000 $W > $O/r02_c2_infer_bf16_pmc_WRITE_SIZE_tail.csv
  head -1 $F > $O/r02_pmc_header.csv
  unset FS_ENGINE_PLAN
  cd $R
fi
if [ $what = all ] || [ $what = steps ]; then
  bash tools/prof_step.sh c3 3 r02_c3_supernet_pretrain_bf16 2>&1 | head -1
  bash tools/prof_step.sh c5 3 r02_c5_supernet_search_bf16 2>&1 | head -1
  bash tools/prof_step.sh c4 5 r02_c4_student_train_bf16 2>&1 | head -1
fi
