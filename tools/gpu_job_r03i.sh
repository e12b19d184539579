#!/bin/bash
# round 3, last GPU call: plan-aligned HBM traffic / MFMA busy of the C2 frame for the TUNED plan (tuned and replayed on one box), and the
# rocprofv3 table of the C3 step issued eagerly (the configuration of bench.py's census step)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export FS_ENGINE_PLAN=$O/r03_c2_plan_choices.json
rm -f $FS_ENGINE_PLAN.*
timeout 300 python bench.py --workloads c2 --no-cpu-baseline --no-class-map --dump-plan $O/r03_c2_plan_inframe_bf16.json > $O/r03_bench_c2_planned.json 2>/dev/null
python tools/extract_c2.py $O/r03_bench_c2_planned.json 2>/dev/null | head -2
cd /tmp
pmc() {
  name=$1; shift; ctr=""
  while [ "$1" != "--" ]; do ctr="$ctr $1"; shift; done; shift
  rm -rf /tmp/pmc_$name
  timeout 150 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$name -o run -- "$@" > $O/r03_pmc_$name.log 2>&1
  find /tmp/pmc_$name -name "*counter_collection.csv" | head -1
}
F=$(pmc c2f FETCH_SIZE -- python $R/tools/profile_frame.py 20 $O/r03_c2_plan_bf16.json)
W=$(pmc c2w WRITE_SIZE -- python $R/tools/profile_frame.py 20)
M=$(pmc c2m SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python $R/tools/profile_frame.py 20)
python $R/tools/pmc_frame.py $O/r03_c2_plan_bf16.json $O/r03_c2_pmc_frame.json fetch=$F write=$W mfma=$M
rm -rf /tmp/prof_c2
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c2 -o run -- python $R/tools/profile_frame.py 60 > $O/r03_prof_c2.log 2>&1
T=$(find /tmp/prof_c2 -name "*kernel_trace.csv" | head -1)
python $R/tools/roofline_from_profile.py frame $O/r03_c2_plan_bf16.json $T $O/r03_bench_c2_planned.json | tee $O/r03_c2_roofline_from_profile.txt
python $R/tools/frame_timeline.py $T $O/r03_c2_infer_bf16_frame_timeline.csv | head -1
unset FS_ENGINE_PLAN
cd $R
FS_SUPERNET_GRAPHS=0 bash tools/prof_step.sh c3 3 r03_c3_supernet_pretrain_bf16_eager 2>&1 | head -1
