#!/bin/bash
# A round's evidence on one MI355X (through gpurun from the repo root): rocprofv3 kernel tables of the C2 frame (plan-order trace) and of
# the C3 / C4 / C5 steps, PMC passes (HBM traffic, MFMA busy) for the C2 frame and the C3 / C4 steps.  Everything lands in gpurun_out/ as
# <tag>_*; the summaries are copied to profiles/ afterwards.  (One script for every round: rounds 2-4 had a copy each.)
# Usage: bash tools/prof_round.sh [c2|steps|pmc_c2|pmc_c3|pmc_c5|pmc_c4|pmc|all] [tag, default r05]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
mkdir -p $O
what=${1:-all}
TAG=${2:-r06}
export FS_ENGINE_PLAN=$O/${TAG}_c2_plan_choices.json
if [ $what = all ] || [ $what = c2 ]; then
  rm -f $FS_ENGINE_PLAN.*
  # tune the plan once on the idle device (the bench's own C2 line), then replay exactly that plan under the profiler
  timeout 400 python bench.py --workloads c2 --no-cpu-baseline --no-class-map --dump-plan $O/${TAG}_c2_plan_inframe_bf16.json > $O/${TAG}_bench_c2_planned.json 2>/dev/null
  python tools/extract_c2.py $O/${TAG}_bench_c2_planned.json 2>/dev/null | head -3
  cd /tmp; rm -rf /tmp/prof_c2
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -o run -- python $R/tools/profile_frame.py 60 $O/${TAG}_c2_plan_bf16.json > $O/${TAG}_prof_c2.log 2>&1
  T=$(find /tmp/prof_c2 -name "*kernel_trace.csv" | head -1)
  cp $(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1) $O/${TAG}_c2_infer_bf16_kernel_stats.csv
  python $R/tools/frame_timeline.py $T $O/${TAG}_c2_infer_bf16_frame_timeline.csv | head -2
  python $R/tools/roofline_from_profile.py frame $O/${TAG}_c2_plan_bf16.json $T $O/${TAG}_bench_c2_planned.json | tee $O/${TAG}_c2_roofline_from_profile.txt
  cd $R
fi
if [ $what = all ] || [ $what = steps ]; then
  bash tools/prof_step.sh c3 3 ${TAG}_c3_supernet_pretrain_bf16 2>&1 | head -1
  bash tools/prof_step.sh c5 3 ${TAG}_c5_supernet_search_bf16 2>&1 | head -1
  bash tools/prof_step.sh c4 5 ${TAG}_c4_student_train_bf16 2>&1 | head -1
  FS_DTYPE=fp32 bash tools/prof_step.sh c3 3 ${TAG}_c3_supernet_pretrain_fp32 2>&1 | head -1
  FS_DTYPE=fp32 bash tools/prof_step.sh c5 3 ${TAG}_c5_supernet_search_fp32 2>&1 | head -1
  FS_DTYPE=fp32 bash tools/prof_step.sh c4 5 ${TAG}_c4_student_train_fp32 2>&1 | head -1
fi
pmc() {  # name counters... -- command...
  name=$1; shift; ctr=""
  while [ "$1" != "--" ]; do ctr="$ctr $1"; shift; done; shift
  rm -rf /tmp/pmc_$name
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$name -o run -- "$@" > $O/${TAG}_pmc_$name.log 2>&1 )
  find /tmp/pmc_$name -name "*counter_collection.csv" | head -1
}
if [ $what = all ] || [ $what = pmc ] || [ $what = pmc_c2 ]; then
  # the frame the plan file describes, issued in plan order: per-kernel table + per-family traffic of the plan (tools/pmc_frame.py)
  F=$(pmc c2f FETCH_SIZE -- python $R/tools/profile_frame.py 20 $O/${TAG}_c2_plan_bf16.json)
  W=$(pmc c2w WRITE_SIZE -- python $R/tools/profile_frame.py 20 $O/${TAG}_c2_plan_bf16.json)
  M=$(pmc c2m SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python $R/tools/profile_frame.py 20 $O/${TAG}_c2_plan_bf16.json)
  python $R/tools/pmc_table.py $O/${TAG}_c2_pmc.json f=$F w=$W m=$M | head -12
  python $R/tools/pmc_frame.py $O/${TAG}_c2_plan_bf16.json $O/${TAG}_c2_pmc_frame.json fetch=$F write=$W mfma=$M | head -12
fi
for wl in c3 c5 c4; do
  if [ $what = all ] || [ $what = pmc ] || [ $what = pmc_$wl ]; then
    # ONE eager step per pass (no capture warm-ups, no graph replays: under --pmc every dispatch is serialised and a pass over a graphed
    # run does not fit the GPU budget); the kernels are the ones the timed steps launch (tests/test_measurement_artifacts.py checks the names)
    # (round 6: the train workloads' printed value is the fp32 step - the traffic tables are taken in fp32 as well)
    export FS_SUPERNET_GRAPHS=0 FS_PROFILE_WARMUP=0 FS_DTYPE=fp32
    F=$(pmc ${wl}f FETCH_SIZE -- python $R/tools/profile_step.py $wl 1)
    W=$(pmc ${wl}w WRITE_SIZE -- python $R/tools/profile_step.py $wl 1)
    if [ $wl = c3 ]; then            # MFMA busy cycles: one workload is enough to cross-check the FLOP-derived fractions
      M=$(pmc ${wl}m SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python $R/tools/profile_step.py $wl 1)
      python $R/tools/pmc_table.py $O/${TAG}_${wl}_pmc.json f=$F w=$W m=$M | head -10
    else
      python $R/tools/pmc_table.py $O/${TAG}_${wl}_pmc.json f=$F w=$W | head -8
    fi
    unset FS_SUPERNET_GRAPHS FS_PROFILE_WARMUP FS_DTYPE
  fi
done
