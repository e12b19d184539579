#!/bin/bash
# Round-4 evidence on one MI355X (through gpurun from the repo root): rocprofv3 kernel tables of the C2 frame (plan-order trace) and of
# the C3 / C4 / C5 steps, PMC passes (HBM traffic, MFMA busy) for C2, C3 and C4.  Everything lands in gpurun_out/; summaries are copied
# to profiles/ afterwards.  Usage: bash tools/prof_r03.sh [c2|steps|pmc|all]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
mkdir -p $O
what=${1:-all}
export FS_ENGINE_PLAN=$O/r04_c2_plan_choices.json
if [ $what = all ] || [ $what = c2 ]; then
  rm -f $FS_ENGINE_PLAN.*
  # tune the plan once on the idle device (the bench's own C2 line), then replay exactly that plan under the profiler
  timeout 400 python bench.py --workloads c2 --no-cpu-baseline --no-class-map --dump-plan $O/r04_c2_plan_inframe_bf16.json > $O/r04_bench_c2_planned.json 2>/dev/null
  python tools/extract_c2.py $O/r04_bench_c2_planned.json 2>/dev/null | head -3
  cd /tmp; rm -rf /tmp/prof_c2
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -o run -- python $R/tools/profile_frame.py 60 $O/r04_c2_plan_bf16.json > $O/r04_prof_c2.log 2>&1
  T=$(find /tmp/prof_c2 -name "*kernel_trace.csv" | head -1)
  cp $(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1) $O/r04_c2_infer_bf16_kernel_stats.csv
  python $R/tools/frame_timeline.py $T $O/r04_c2_infer_bf16_frame_timeline.csv | head -2
  python $R/tools/roofline_from_profile.py frame $O/r04_c2_plan_bf16.json $T $O/r04_bench_c2_planned.json | tee $O/r04_c2_roofline_from_profile.txt
  cd $R
fi
if [ $what = all ] || [ $what = steps ]; then
  bash tools/prof_step.sh c3 3 r04_c3_supernet_pretrain_bf16 2>&1 | head -1
  bash tools/prof_step.sh c5 3 r04_c5_supernet_search_bf16 2>&1 | head -1
  bash tools/prof_step.sh c4 5 r04_c4_student_train_bf16 2>&1 | head -1
  FS_DTYPE=fp32 bash tools/prof_step.sh c3 3 r04_c3_supernet_pretrain_fp32 2>&1 | head -1
fi
if [ $what = all ] || [ $what = pmc ]; then
  cd /tmp
  pmc() {  # name counters... -- command...
    name=$1; shift; ctr=""
    while [ "$1" != "--" ]; do ctr="$ctr $1"; shift; done; shift
    rm -rf /tmp/pmc_$name
    timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$name -o run -- "$@" > $O/r04_pmc_$name.log 2>&1
    find /tmp/pmc_$name -name "*counter_collection.csv" | head -1
  }
  F=$(pmc c2f FETCH_SIZE -- python $R/tools/profile_frame.py 20)
  W=$(pmc c2w WRITE_SIZE -- python $R/tools/profile_frame.py 20)
  M=$(pmc c2m SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python $R/tools/profile_frame.py 20)
  python $R/tools/pmc_table.py $O/r04_c2_pmc.json f=$F w=$W m=$M | head -12
  python $R/tools/pmc_traffic.py $F $W bf16 $O/r04_pmc_traffic.json > /dev/null 2>&1
  for wl in c3 c4; do
    F=$(pmc ${wl}f FETCH_SIZE -- python $R/tools/profile_step.py $wl 2)
    W=$(pmc ${wl}w WRITE_SIZE -- python $R/tools/profile_step.py $wl 2)
    M=$(pmc ${wl}m SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python $R/tools/profile_step.py $wl 2)
    B=$(grep -o "PROFILE_STEPS_BEGIN [0-9]* [0-9]*" $O/r04_pmc_${wl}m.log | awk '{print $3}'); E=$(grep -o "PROFILE_STEPS_END [0-9]* [0-9]*" $O/r04_pmc_${wl}m.log | awk '{print $3}')
    python $R/tools/pmc_table.py $O/r04_${wl}_pmc.json f=$F w=$W m=$M | head -10
  done
  cd $R
fi
