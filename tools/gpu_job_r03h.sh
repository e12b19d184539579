#!/bin/bash
# round 3, eighth GPU call: the prewarm fix, plan-aligned HBM traffic of the C2 frame (conv3x3 / conv1x1 separately), final default bench line
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_train_steps_gpu.py -q -x > $O/r03h_steps.log 2>&1; tail -3 $O/r03h_steps.log
cd /tmp
export FS_ENGINE_PLAN=$O/r03_c2_plan_choices.json
pmc() {
  name=$1; shift; ctr=""
  while [ "$1" != "--" ]; do ctr="$ctr $1"; shift; done; shift
  rm -rf /tmp/pmc_$name
  timeout 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$name -o run -- "$@" > $O/r03_pmc_$name.log 2>&1
  find /tmp/pmc_$name -name "*counter_collection.csv" | head -1
}
F=$(pmc c2f FETCH_SIZE -- python $R/tools/profile_frame.py 20 $O/r03_c2_plan_bf16_pmc.json)
W=$(pmc c2w WRITE_SIZE -- python $R/tools/profile_frame.py 20)
M=$(pmc c2m SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python $R/tools/profile_frame.py 20)
python $R/tools/pmc_frame.py $O/r03_c2_plan_bf16_pmc.json $O/r03_c2_pmc_frame.json fetch=$F write=$W mfma=$M
cp $O/r03_c2_pmc_frame.json $R/profiles/r03_c2_pmc_frame.json
unset FS_ENGINE_PLAN
cd $R
timeout 1200 python bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err; tail -c 300 $O/r03_bench_default.err
python tools/extract_bench.py $O/r03_bench_default.json
